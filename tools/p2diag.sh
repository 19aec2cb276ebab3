# timing of the patch-kernel forms on the trunk shapes (tune library: knobs from the environment)
T=$PWD/unsupervised-pseuso-lidar_amd/mcav/libmcav_depth_tune.so
run() { echo "== $1"; shift; env "$@" MCAV_LIB_PATH=$T CONV_BENCH_MMA=2 CONV_BENCH_BATCH=24 CONV_BENCH_SHAPES=0,1,2,3 python tools/conv_bench.py ${WHAT:-fwd} 2>&1 | grep -v amdgpu.ids; }
run "v1 (MCAV_PATCH2=0)" MCAV_PATCH2=0
run "v2 auto" MCAV_PATCH2=1
run "v2 SB=2 BN=64" MCAV_PATCH2_SB=2
run "v2 SB=4 BN=64" MCAV_PATCH2_SB=4
run "v2 SB=2 BN=32 (2 WG/CU)" MCAV_PATCH2_BN=32
WHAT=dgrad run "dgrad v2 SB=2 BN=32" MCAV_PATCH2_BN=32
