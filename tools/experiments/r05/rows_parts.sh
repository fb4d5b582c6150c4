# composition of a fneus_mlp_* launch: parts compiled out one at a time (times only; results are wrong)
python3 tools/experiments/r05/mlp_rows_time.py 2>&1 | head -3
for v in rows_empty rows_nomfma rows_noloads; do echo "== $v"; FNEUS_LIB=$PWD/factored-neus_amd/fneus/variants/libfneus_$v.so python3 tools/experiments/r05/mlp_rows_time.py 2>&1 | head -3; done
