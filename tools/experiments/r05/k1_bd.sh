# B-fragment prefetch distance of the two-pass K1 (FNEUS_P2_BD): 1 = shipped
python3 tools/experiments/r05/k1_h6_time.py 2>&1 | grep "seed 20 n 65536" | sed 's/h6.*//'
for v in p2_bd2 p2_bd3; do echo "== $v"; FNEUS_LIB=$PWD/factored-neus_amd/fneus/variants/libfneus_$v.so python3 tools/experiments/r05/k1_h6_time.py 2>&1 | grep "seed 20 n 65536" | sed 's/h6.*//'; done
