for n in ${ABLS:-1 2 3 4 5}; do echo "ABL $n"; DI2P_LIB=$PWD/deepi2p_amd/lib/variants/abl$n/libdeepi2p_hip.so timeout 100 python tools/bench_stem_x3.py 2>&1 | grep "one launch"; done
