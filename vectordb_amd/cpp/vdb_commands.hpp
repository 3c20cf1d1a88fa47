// vdb_commands.hpp — the vector commands of SURVEY.md §8f row N4: what a maintainer of the reference adds to its
// server so that clients reach the index (`VADD`, `VSEARCH`, `VDEL`, `VCARD`), written against the reference's own
// dispatch: handlers take (args, response bytes) as CommandProcessor::CommandContext does (command_processor.hpp:29-33)
// and answer through the static functions of its ResponseSerializer (response_serializer.hpp:31-58), passed as a
// template parameter so that this header needs none of the server's sources.  In the reference:
//
//     static vdb_amd::VectorCommands<ResponseSerializer> vectors;                      // beside command_handlers
//     {"vadd",    [](CommandContext c) { vectors.vadd(c.args, c.response); }},         // command_processor.hpp:237-248
//     {"vsearch", [](CommandContext c) { vectors.vsearch(c.args, c.response); }}, ...
//
// CPU glue only: parsing, the per-key index table, the reply.  The search itself is the device path of vdb_hnsw.hpp.
#pragma once
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "vdb_hnsw.hpp"

namespace vdb_amd {

template <typename Serializer>
class VectorCommands {
public:
    // the reference's codes, command_processor.hpp:20-22
    static constexpr int kErrArg = -1, kErrUnknown = -2, kErrType = -3;
    size_t M = 16, ef_construction = 200, initial_capacity = 1024;
    int device = 0;

    // VADD key label f32 x dim      -> integer 1.  The first VADD of a key fixes its dimension (L2 distance).
    void vadd(const std::vector<std::string>& args, std::vector<uint8_t>& out) {
        if (args.size() < 4) return Serializer::serialize_error(out, kErrArg, "VADD requires key, label and a vector\n");
        long long label;
        std::vector<float> v;
        if (!parse_int(args[2], label) || label < 0) return Serializer::serialize_error(out, kErrArg, "Invalid label\n");
        if (!parse_vector(args, 3, args.size(), v)) return Serializer::serialize_error(out, kErrArg, "Invalid vector component\n");
        std::unique_lock<std::mutex> g(lock_);
        Slot* s = find(args[1]);
        if (!s) {
            // the key enters the table only once its index exists: a constructor that throws (no device memory, a
            // dimension the device path refuses) leaves no half-made slot behind for later commands to trip over
            try {
                s = &create(args[1], v.size());
            } catch (const std::exception& e) {
                return Serializer::serialize_error(out, kErrUnknown, std::string(e.what()) + "\n");
            }
        }
        if (s->dim != v.size()) return Serializer::serialize_error(out, kErrArg, "Vector has the wrong dimension for this key\n");
        try {
            if (s->index->getElementCount() == s->capacity) {  // grow as resizeIndex does (hnsw_core.hpp:500-522)
                s->capacity *= 2;
                s->index->resizeIndex(s->capacity);
            }
            s->index->addPoint(v.data(), (labeltype)label);
        } catch (const std::exception& e) {
            return Serializer::serialize_error(out, kErrUnknown, std::string(e.what()) + "\n");
        }
        Serializer::serialize_integer(out, 1);
    }

    // VSEARCH key k f32 x dim [EF ef] -> integer n, then n x (integer label, double distance), closest first
    void vsearch(const std::vector<std::string>& args, std::vector<uint8_t>& out) {
        if (args.size() < 4) return Serializer::serialize_error(out, kErrArg, "VSEARCH requires key, k and a vector\n");
        long long k, ef = 0;
        if (!parse_int(args[2], k) || k <= 0) return Serializer::serialize_error(out, kErrArg, "Invalid k\n");
        size_t end = args.size();
        if (end >= 6 && (args[end - 2] == "EF" || args[end - 2] == "ef")) {
            if (!parse_int(args[end - 1], ef) || ef <= 0) return Serializer::serialize_error(out, kErrArg, "Invalid EF\n");
            end -= 2;
        }
        std::vector<float> q;
        if (!parse_vector(args, 3, end, q)) return Serializer::serialize_error(out, kErrArg, "Invalid vector component\n");
        std::unique_lock<std::mutex> g(lock_);
        Slot* s = find(args[1]);
        if (!s) return Serializer::serialize_integer(out, 0);  // no such key: no neighbours
        if (s->dim != q.size()) return Serializer::serialize_error(out, kErrArg, "Vector has the wrong dimension for this key\n");
        std::vector<std::pair<float, labeltype>> res;
        try {
            if (ef) s->index->setefSearch((size_t)ef);
            res = s->index->SearchKNNCloserFirst(q.data(), (size_t)k);
        } catch (const std::exception& e) {
            return Serializer::serialize_error(out, kErrUnknown, std::string(e.what()) + "\n");
        }
        Serializer::serialize_integer(out, (int64_t)res.size());
        for (const auto& p : res) {
            Serializer::serialize_integer(out, (int64_t)p.second);
            Serializer::serialize_double(out, (double)p.first);
        }
    }

    // VDEL key label -> integer 1 (marked deleted: markDelete, hnsw_core.hpp:709) or 0 (no such key / label)
    void vdel(const std::vector<std::string>& args, std::vector<uint8_t>& out) {
        if (args.size() != 3) return Serializer::serialize_error(out, kErrArg, "VDEL requires key and label\n");
        long long label;
        if (!parse_int(args[2], label) || label < 0) return Serializer::serialize_error(out, kErrArg, "Invalid label\n");
        std::unique_lock<std::mutex> g(lock_);
        Slot* s = find(args[1]);
        if (!s) return Serializer::serialize_integer(out, 0);
        try {
            s->index->markDelete((labeltype)label);
        } catch (const std::exception&) {
            return Serializer::serialize_integer(out, 0);
        }
        Serializer::serialize_integer(out, 1);
    }

    // VCARD key -> integer: elements stored under the key
    void vcard(const std::vector<std::string>& args, std::vector<uint8_t>& out) {
        if (args.size() != 2) return Serializer::serialize_error(out, kErrArg, "VCARD requires one key\n");
        std::unique_lock<std::mutex> g(lock_);
        Slot* s = find(args[1]);
        Serializer::serialize_integer(out, s ? (int64_t)s->index->getElementCount() : 0);
    }

    // the index behind a key (nullptr: no such key), for the embedding server's own use (persistence, statistics)
    HierarchicalNSW<float>* index(const std::string& key) {
        std::unique_lock<std::mutex> g(lock_);
        Slot* s = find(key);
        return s ? s->index.get() : nullptr;
    }

private:
    struct Slot {
        size_t dim = 0, capacity = 0;
        std::unique_ptr<L2Space<float>> space;
        std::unique_ptr<HierarchicalNSW<float>> index;
    };
    std::mutex lock_;
    std::map<std::string, Slot> table_;

    Slot* find(const std::string& key) {
        auto it = table_.find(key);
        return it == table_.end() ? nullptr : &it->second;
    }
    Slot& create(const std::string& key, size_t dim) {
        Slot s;
        s.dim = dim;
        s.capacity = initial_capacity;
        s.space.reset(new L2Space<float>(dim));
        s.index.reset(new HierarchicalNSW<float>(s.space.get(), s.capacity, M, ef_construction, 100, false, device));  // may throw
        return table_[key] = std::move(s);
    }
    static bool parse_int(const std::string& t, long long& v) {
        errno = 0;
        char* e = nullptr;
        v = std::strtoll(t.c_str(), &e, 10);
        return errno == 0 && e != t.c_str() && *e == '\0';
    }
    static bool parse_vector(const std::vector<std::string>& args, size_t lo, size_t hi, std::vector<float>& v) {
        v.clear();
        for (size_t i = lo; i < hi; ++i) {
            errno = 0;
            char* e = nullptr;
            const float f = std::strtof(args[i].c_str(), &e);
            if (errno != 0 || e == args[i].c_str() || *e != '\0' || !std::isfinite(f)) return false;
            v.push_back(f);
        }
        return !v.empty();
    }
};

}  // namespace vdb_amd
