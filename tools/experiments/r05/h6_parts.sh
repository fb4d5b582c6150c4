# composition of the h6 K1 launch: vector work, MFMAs, weight stream compiled out one at a time (results are then wrong: times only)
python3 tools/experiments/r05/k1_h6_time.py 2>&1 | grep "net seed"
for v in gs4 nomfma; do echo "== $v"; FNEUS_LIB=$PWD/factored-neus_amd/fneus/variants/libfneus_$v.so python3 tools/experiments/r05/k1_h6_time.py 2>&1 | grep "seed 20 n 65536" | sed 's/bf1.*//'; done
