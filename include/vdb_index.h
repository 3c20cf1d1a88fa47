/* vdb_index.h — C-ABI of the MI355X-native HNSW / brute-force k-NN path.
 *
 * This is the drop-in boundary for ONE hot path of cheeriosok/VectorDB: the k-NN query of
 * src/hnsw/hnsw_scratch (greedy upper-layer descent + ef-bounded layer-0 beam search) and
 * the exact brute-force k-NN beside it.  Each entry point names the reference interface
 * it stands in for (paths relative to the reference's src/hnsw/hnsw_scratch/).
 * The reference surface is C++ (exceptions, std::priority_queue results); across this ABI
 * everything is plain pointers and sizes, every function returns a status code
 * (0 = VDB_OK) and vdb_last_error() gives the thread-local message that the C++ wrapper
 * (vectordb_amd/cpp/vdb_hnsw.hpp) re-throws as std::runtime_error, as the reference does.
 *
 * Host-side index state keeps the reference's byte layout (hnsw_core.hpp:45-75, :148-152)
 * so saveIndex/loadIndex files are interchangeable; vdb_index_to_device repacks it into
 * the HBM layout described in DESIGN.md.  Search entry points run hand-written gfx950
 * kernels only: there is no CPU fallback, and they fail with VDB_ERR_NO_DEVICE when the
 * index has not been uploaded or no GPU is present.
 */
#ifndef VDB_INDEX_H
#define VDB_INDEX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vdb_index vdb_index;           /* HierarchicalNSW<float>, hnsw_core.hpp:21 */
typedef struct vdb_bruteforce vdb_bruteforce; /* BruteforceSearch<float>, bruteforce.hpp:14 */

enum vdb_metric {
    VDB_L2 = 0, /* L2Sqr<float>, space_l2.hpp:1-14 */
    VDB_IP = 1  /* InnerProductDistance<float> = 1 - <a,b>, space_ip.hpp:13-18 */
};

enum vdb_status {
    VDB_OK = 0,
    VDB_ERR_INVALID = 1,     /* bad argument */
    VDB_ERR_RUNTIME = 2,     /* what the reference reports as std::runtime_error */
    VDB_ERR_NO_DEVICE = 3,   /* no GPU / index not on device / HIP call failed */
    VDB_ERR_UNSUPPORTED = 4  /* valid in the reference, not on the device path yet */
};

/* Order in which a distance is summed (DESIGN.md "Numerics").
 * VDB_SUM_TREE64 is what the kernels compute and the default for building;
 * VDB_SUM_SEQ is the reference's strictly sequential fp32 loop (host build only). */
enum vdb_sum_order { VDB_SUM_SEQ = 0, VDB_SUM_TREE64 = 1 };

/* Precision of the brute-force distance engine. */
enum vdb_bf_precision {
    VDB_BF_FP32 = 0,       /* f32 MFMA candidate pass + exact f32 re-rank: ids/dists match the CPU path */
    VDB_BF_BF16 = 1,       /* bf16 MFMA, f32 accumulate, no re-rank: distances are bf16-grade */
    VDB_BF_BF16_RERANK = 2 /* bf16 MFMA candidate pass (wider margin) + exact f32 re-rank */
};

const char* vdb_last_error(void);
/* Number of visible HIP devices (0 when none; never fails). */
int vdb_device_count(void);

/* ---- HierarchicalNSW<float> ------------------------------------------------------- */

/* ctor HierarchicalNSW(space, capacity, M=16, efConstruction=200, random_seed=100),
 * hnsw_core.hpp:118-179; `metric`+`dim` replace the SpaceInterface argument
 * (hnswlib.hpp:54-61, space_l2.hpp:16-37, space_ip.hpp:20-41). */
int vdb_index_create(vdb_index** out, int metric, size_t dim, size_t capacity, size_t M,
                     size_t ef_construction, size_t random_seed);
/* load ctor / loadIndex, hnsw_core.hpp:110-116, :581-682 (reference byte format). */
int vdb_index_load(vdb_index** out, const char* path, int metric, size_t dim, size_t capacity);
/* saveIndex, hnsw_core.hpp:551-579. */
int vdb_index_save(const vdb_index* idx, const char* path);
/* ~HierarchicalNSW / clear(), hnsw_core.hpp:181-199 (also frees the device copy). */
void vdb_index_free(vdb_index* idx);

/* addPoint(datapoint, label), hnsw_core.hpp:804-842 -> :994-1104.  Host build with the
 * reference's semantics (same graph as the reference algorithm for the same insertion
 * order, seed and summation order).  `level` < 0 draws the level (hnsw_core.hpp:240-244). */
int vdb_index_add(vdb_index* idx, const float* vec, uint64_t label, int level);
/* addPoint(datapoint, label, replace_deleted), hnsw_core.hpp:804-842.  A label the index holds already is UPDATED
 * in place (hnsw_core.hpp:1001-1016 -> updatePoint :844-913 -> repairConnectionsForUpdate :924-980; a deleted
 * element comes back to life); vdb_index_add does the same.  With replace_deleted != 0 the element takes the place
 * of a deleted one when there is any (:815-841: the smallest vacant internal id; the reference takes an unspecified
 * one) and is appended otherwise; VDB_ERR_RUNTIME "Replacement of deleted elements is disabled in constructor"
 * unless vdb_index_set_reuse_deleted switched it on (:805-807). */
int vdb_index_add_point(vdb_index* idx, const float* vec, uint64_t label, int replace_deleted);
/* The constructors' reuse_deleted flag, hnsw_core.hpp:114, :123 (and :672-677: elements already marked deleted
 * become vacant places).  With it on, vdb_index_add on the label of a DELETED element is refused (:1003-1007). */
int vdb_index_set_reuse_deleted(vdb_index* idx, int on);
/* n addPoint calls; labels NULL = 0..n-1 continuing from the current count.  n_threads > 1
 * inserts concurrently as the reference allows (label stripes, per-node link locks,
 * global lock: hnsw_core.hpp:77-84); the link order is then not reproducible. */
int vdb_index_add_batch(vdb_index* idx, const float* vecs, const uint64_t* labels, size_t n, int n_threads);
/* The same insertion with layer 0 built on `device` (no reference counterpart: the
 * reference builds on host threads only, hnsw_core.hpp:994-1104).  Elements that have upper levels,
 * and the first 2048 of a new graph, are inserted by the host builder above; the others in lock-free
 * rounds of search (ef_construction) -> heuristic selection (:357-390) -> reverse links (:392-498) on
 * the GPU.  Levels, ids and labels are assigned exactly as vdb_index_add_batch does; the links are a
 * valid HNSW graph but not the sequential builder's graph (a batch has no insertion order).  Leaves
 * the index uploaded to `device`.  Needs ef_construction <= 508, M <= 32, no deleted elements.
 * On failure an index that was empty is empty again; for an append the new elements stay and the ones
 * the device did not link are linked by the host builder before the error is returned. */
int vdb_index_add_batch_device(vdb_index* idx, const float* vecs, const uint64_t* labels, size_t n, int n_threads,
                               int device);
/* Summation order used by host-side build distances (default VDB_SUM_TREE64). */
int vdb_index_set_build_sum_order(vdb_index* idx, int order);

/* setefSearch, hnsw_core.hpp:207-209 (default 10, :143/:655). */
int vdb_index_set_ef(vdb_index* idx, size_t ef);
/* Tuning switch of the device search (no reference counterpart; the reference's visited list is an O(n)
 * array, visited_list_pool.hpp:36-58): whether queries that fill their LDS visited table carry on in a
 * second-level table in HBM.  -1 = decided per ef from what earlier launches reported (default),
 * 0 = never, 1 = always.  Results are identical in every mode.  The default is taken from the
 * environment variable VDB_SPILL (0 / 1) once, when the index is created. */
int vdb_index_set_spill_mode(vdb_index* idx, int mode);
/* markDelete / unmarkDelete, hnsw_core.hpp:709-782. */
int vdb_index_mark_delete(vdb_index* idx, uint64_t label);
int vdb_index_unmark_delete(vdb_index* idx, uint64_t label);
/* getDataByLabel, hnsw_core.hpp:683-703: the stored vector (dim floats); "Label not found" for an unknown or a
 * deleted label. */
int vdb_index_get_data_by_label(const vdb_index* idx, uint64_t label, float* out);
/* resizeIndex, hnsw_core.hpp:500-522. */
int vdb_index_resize(vdb_index* idx, size_t new_capacity);
/* checkIntegrity, hnsw_core.hpp:1216-1245: 0 ok, else 1 id out of range, 2 self link,
 * 3 duplicate link, 4 node without inbound link. */
int vdb_index_check_integrity(const vdb_index* idx, int* min_inbound, int* max_inbound);

/* getCapacity / getElementCount / getDeletedCount (hnsw_core.hpp:246-256) and the public
 * fields the reference exposes (:28-40). */
size_t vdb_index_capacity(const vdb_index* idx);
size_t vdb_index_count(const vdb_index* idx);
size_t vdb_index_deleted_count(const vdb_index* idx);
size_t vdb_index_dim(const vdb_index* idx);
size_t vdb_index_M(const vdb_index* idx);
size_t vdb_index_ef(const vdb_index* idx);
int vdb_index_max_level(const vdb_index* idx);
uint32_t vdb_index_entry(const vdb_index* idx);
int vdb_index_level_of(const vdb_index* idx, uint32_t internal_id);
/* One node's neighbour list on one level, in stored order (get_neighbors_at_level,
 * hnsw_core.hpp:259-273); returns the count or -1. */
int vdb_index_links(const vdb_index* idx, uint32_t internal_id, int level, uint32_t* out, int cap);

/* Repack the host graph into the HBM layout and upload it to `device` (no reference
 * counterpart: the reference has no device).  Must be called again after add/delete. */
int vdb_index_to_device(vdb_index* idx, int device);
/* loadIndex (hnsw_core.hpp:581-682) followed by vdb_index_to_device, without the host copy of the level-0 block
 * in between: the header, the levels and the upper-layer lists (a few percent of the file) are read first, then
 * the rows [list | vector | label] stream from the file in 64 Ki-row chunks through the same repack, validation
 * and upload as vdb_index_to_device (identical device image; a 65 GB index needs ~100 MB of host memory, not
 * 65 GB).  The index it returns searches (every vdb_search_* entry point, efSearch, stats) and can be
 * uploaded again; what needs the rows on the host - add, markDelete, save, resize, links of layer 0, labels,
 * integrity check - fails with VDB_ERR_RUNTIME and says so: use vdb_index_load for those. */
int vdb_index_load_to_device(vdb_index** out, const char* path, int metric, size_t dim, int device);
/* Bytes the device copy occupies (0 if not uploaded). */
size_t vdb_index_device_bytes(const vdb_index* idx);

/* SearchKNNCloserFirst over a batch: hnswlib.hpp:87-99 on top of searchKnn,
 * hnsw_core.hpp:1107-1159 (+ the undefined searchBaseLayerST it calls, :1143-1147).
 * One wavefront per query.  ef = max(efSearch, k) as :1144.
 *   d_queries  [nq * dim] f32, DEVICE memory, row-major
 *   d_out_dist [nq * k]   f32, DEVICE, ascending; rows short of k padded with +inf
 *   d_out_label[nq * k]   u64, DEVICE; padding = UINT64_MAX
 *   d_out_stats nullable  [nq * 4] u32, DEVICE: {ndis, nhops layer 0, nhops upper, found}
 *                         (the reference's metric_distance_computations_/metric_hops_,
 *                          hnsw_core.hpp:95-96, :1122-1123, per query)
 *   stream     hipStream_t (NULL = default stream); the call is asynchronous.
 * Elements marked deleted before the upload are skipped as the reference's non-bare path does
 * (:1141-1148: they are traversed, never returned). */
int vdb_search_batch_device(const vdb_index* idx, const float* d_queries, size_t nq, size_t k,
                            float* d_out_dist, uint64_t* d_out_label, uint32_t* d_out_stats, void* stream);
/* vdb_search_batch_device for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference (it has no
 * device).  Same arguments, same results bit for bit, and it calls no hipMalloc / hipFree / allocating
 * library routine: the traversal's scratch (the work list of queries that outgrow their LDS visited table
 * and the visited bitmaps of the waves that re-run them) lives in a static device array of the library,
 * about 4.1 MB per GPU, held until the process ends.  Limits (VDB_ERR_INVALID beyond them): nq <= 32768
 * queries per call, index of at most 2^23 elements.  No caller's filter; deleted elements are skipped as
 * above.  A query whose key array overflows (hundreds of equal distances) is reported as described at
 * vdb_search_batch_device_filtered.  Every call on a GPU shares that GPU's array: calls that overlap
 * on one GPU (two streams, two threads) race - serialise them. */
int vdb_search_batch_device_noalloc(const vdb_index* idx, const float* d_queries, size_t nq, size_t k,
                                    float* d_out_dist, uint64_t* d_out_label, uint32_t* d_out_stats, void* stream);
/* Same with HOST pointers (copies in and out, synchronises; PCIe-inclusive). */
int vdb_search_batch(const vdb_index* idx, const float* queries, size_t nq, size_t k, float* out_dist,
                     uint64_t* out_label, uint32_t* out_stats);

/* searchKnn(query, k, filter): the non-bare searchBaseLayerST of hnsw_core.hpp:1141-1148 with a
 * BaseFilterFunctor (hnswlib.hpp:10-23).  A functor cannot cross the ABI; its verdicts can:
 *   d_allow  DEVICE bitmap over INTERNAL ids, ceil(count / 32) u32 words, bit (i % 32) of word
 *            i / 32 set = filter(label of element i) is true.  vdb_index_labels() gives the labels
 *            in internal-id order to evaluate the functor on.  NULL = no filter.
 * Deleted elements are excluded as well.  Nodes that are not allowed are still traversed (:340-343).
 * A query whose live candidates outside the result set outnumber what the register configurations keep
 * (256-1024 keys by ef: a filter that rejects almost everything) is run once more with its candidates in
 * a heap in HBM (hnsw_search_unbounded_kernel): answered as the reference's unbounded heaps would, only
 * slowly (brute force over the allowed rows is the faster tool at such selectivity).  What remains
 * reported instead of answered - row NaN / UINT64_MAX, stats.found = UINT32_MAX, VDB_ERR_UNSUPPORTED from
 * the host-pointer variant (the other rows are valid): more than 2^25 candidates at once, and, on the
 * unfiltered path and in vdb_search_batch_device_noalloc, hundreds of base vectors at exactly the
 * distance of the ef-th result. */
int vdb_search_batch_device_filtered(const vdb_index* idx, const float* d_queries, size_t nq, size_t k,
                                     const uint32_t* d_allow, float* d_out_dist, uint64_t* d_out_label,
                                     uint32_t* d_out_stats, void* stream);
int vdb_search_batch_filtered(const vdb_index* idx, const float* queries, size_t nq, size_t k, const uint32_t* allow,
                              float* out_dist, uint64_t* out_label, uint32_t* out_stats);
/* searchStopConditionClosest(query, stop_condition, filter): hnsw_core.hpp:1162-1213 - the greedy descent of
 * searchKnn, then the non-bare layer-0 search driven by a BaseSearchStopCondition (hnswlib.hpp:10-23) instead of
 * ef, results closest first, the condition's filter_results last.  Six host callbacks per candidate cannot run in a
 * kernel, and the reference defines no concrete condition, so ONE crosses the ABI as plain numbers: the epsilon-radius
 * condition of the library the reference's index derives from (hnswlib 0.8.0, EpsilonSearchStopCondition) - stop once
 * the best open candidate lies beyond `epsilon` and min_num_candidates results are held; keep at most
 * max_num_candidates (<= 508) results; return those within epsilon.  Parity of this entry point is unpinned
 * (oracle/vdb_oracle.hpp says why).
 *   d_out_dist / d_out_label  [nq * max_num_candidates], ascending (dist, label), padded with +inf / UINT64_MAX
 *   d_out_stats  nullable [nq * 4] as above; stats[4 q + 3] = number of results of query q
 *   d_allow      as vdb_search_batch_device_filtered; NULL = every element (deleted ones are always excluded)
 * DEVIATION: :1205 hands back INTERNAL ids; this returns labels like every other search. */
int vdb_search_epsilon_batch_device(const vdb_index* idx, const float* d_queries, size_t nq, float epsilon,
                                    size_t min_num_candidates, size_t max_num_candidates, const uint32_t* d_allow,
                                    float* d_out_dist, uint64_t* d_out_label, uint32_t* d_out_stats, void* stream);
int vdb_search_epsilon_batch(const vdb_index* idx, const float* queries, size_t nq, float epsilon, size_t min_num_candidates,
                             size_t max_num_candidates, const uint32_t* allow, float* out_dist, uint64_t* out_label,
                             uint32_t* out_stats);
/* getExternalLabel (hnsw_core.hpp:219-221) for internal ids 0 .. count-1; cap >= count. */
int vdb_index_labels(const vdb_index* idx, uint64_t* out, size_t cap);

/* ---- BruteforceSearch<float> ------------------------------------------------------ */

/* ctor BruteforceSearch(space, maxElements), bruteforce.hpp:54-65. */
int vdb_bf_create(vdb_bruteforce** out, int metric, size_t dim, size_t capacity);
/* loadIndex / saveIndex, bruteforce.hpp:190-231. */
int vdb_bf_load(vdb_bruteforce** out, const char* path, int metric, size_t dim);
int vdb_bf_save(const vdb_bruteforce* bf, const char* path);
void vdb_bf_free(vdb_bruteforce* bf);
/* addPoint / removePoint, bruteforce.hpp:84-138. */
int vdb_bf_add_batch(vdb_bruteforce* bf, const float* vecs, const uint64_t* labels, size_t n);
int vdb_bf_remove(vdb_bruteforce* bf, uint64_t label);
size_t vdb_bf_count(const vdb_bruteforce* bf);
/* Upload rows + labels (+ norms) to `device`. */
int vdb_bf_to_device(vdb_bruteforce* bf, int device);
/* SearchKNN (bruteforce.hpp:147-184) closest first over a batch: the k smallest
 * (dist, label) pairs per query.  Device pointers as vdb_search_batch_device. */
int vdb_bf_search_batch_device(const vdb_bruteforce* bf, const float* d_queries, size_t nq, size_t k,
                               int precision, float* d_out_dist, uint64_t* d_out_label, void* stream);
int vdb_bf_search_batch(const vdb_bruteforce* bf, const float* queries, size_t nq, size_t k, int precision,
                        float* out_dist, uint64_t* out_label);

/* SearchKNN(query, K, filter), bruteforce.hpp:147-184 with a BaseFilterFunctor (hnswlib.hpp:64-68).  The functor's
 * verdicts cross the ABI as a bitmap over ROWS in storage order (vdb_bf_labels lists the labels to call it on):
 * bit (i % 32) of word i / 32 set = accepted; ceil(count / 32) words.  Always the exact scan (results as
 * VDB_BF_FP32).  The reference's filtered scan is order dependent - a row beyond the first K is only considered
 * when its distance is at or below the worst ACCEPTED row of the first K (:163-167) - and that is reproduced. */
int vdb_bf_search_batch_device_filtered(const vdb_bruteforce* bf, const float* d_queries, size_t nq, size_t k,
                                        const uint32_t* d_allow, float* d_out_dist, uint64_t* d_out_label, void* stream);
int vdb_bf_search_batch_filtered(const vdb_bruteforce* bf, const float* queries, size_t nq, size_t k, const uint32_t* allow,
                                 float* out_dist, uint64_t* out_label);
/* The label of every stored row, in storage order (bruteforce.hpp:156-157); cap >= count. */
int vdb_bf_labels(const vdb_bruteforce* bf, uint64_t* out, size_t cap);

/* ---- shard merge (no reference counterpart) --------------------------- */

/* Merge G per-shard result lists into one: inputs [G][nq][k] (each row ascending, padded
 * with +inf/UINT64_MAX), output [nq][k] = the k smallest (dist, label).  All DEVICE. */
int vdb_merge_topk_device(const float* d_dist, const uint64_t* d_label, size_t n_lists, size_t nq, size_t k,
                          float* d_out_dist, uint64_t* d_out_label, void* stream);

/* ---- base-set sharding over the GPUs of a node (no reference counterpart) -----------
 *
 * One process (or thread) per GPU, each holding one shard: an independent HierarchicalNSW over a
 * contiguous range of the base rows whose labels are the global row ids.  Every rank calls
 * vdb_sharded_search_batch_device with the SAME query batch; the per-rank [nq][k] result lists are
 * exchanged with ncclAllGather over RCCL (xGMI inside a node) and merged on every rank.  RCCL is bound at
 * run time (dlopen of librccl.so.1): a process that never shards does not need it. */
typedef struct vdb_comm vdb_comm;
#define VDB_COMM_ID_BYTES 128
/* ncclGetUniqueId on ONE rank; ship the 128 bytes to the others by any means (file, socket, MPI, ...). */
int vdb_comm_unique_id(void* id128);
/* ncclCommInitRank: collective over the n_ranks callers; `device` is the GPU this rank's shard lives on. */
int vdb_comm_create(vdb_comm** out, const void* id128, int n_ranks, int rank, int device);
/* The same around a communicator the caller already has (an ncclComm_t); it is not destroyed by vdb_comm_free. */
int vdb_comm_wrap(vdb_comm** out, void* nccl_comm, int device);
void vdb_comm_free(vdb_comm* comm);
int vdb_comm_size(const vdb_comm* comm);
int vdb_comm_rank(const vdb_comm* comm);
/* searchKnn over the whole sharded base set: this rank's vdb_search_batch_device on `shard`, one grouped
 * ncclAllGather of the [nq][k] distance and label lists, vdb_merge_topk_device - all on `stream`, no host
 * synchronisation.  Outputs as vdb_search_batch_device, identical on every rank.  n_ranks * k <= 5461. */
int vdb_sharded_search_batch_device(const vdb_index* shard, vdb_comm* comm, const float* d_queries, size_t nq, size_t k,
                                    float* d_out_dist, uint64_t* d_out_label, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VDB_INDEX_H */
