set -e
bash tools/prof_c2.sh
EFS="32 100 256" bash tools/prof_hnsw_ef.sh
