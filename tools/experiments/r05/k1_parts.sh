# composition of the shipped two-pass K1 (three bf16 products): parts compiled out one at a time (times only; results are wrong)
for v in p2_nomfma p2_novalu p2_valuonly p2_valuonly_nw p2_mfmaonly; do echo "== $v"; FNEUS_LIB=$PWD/factored-neus_amd/fneus/variants/libfneus_$v.so python3 tools/experiments/r05/k1_h6_time.py 2>&1 | grep "seed 20 n 65536" | sed 's/h6.*//'; done
