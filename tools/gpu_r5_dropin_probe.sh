S=tests/bitstreams/mini_4k_all_tools_ctu128_3840x2176/mini_4k_all_tools_ctu128_3840x2176.bit
R=oracle/_ref
export LD_LIBRARY_PATH=$R:$PWD/vvdec_amd:$LD_LIBRARY_PATH
for bt in 2 4 8 12; do for args in "-t 16" "-t 32"; do echo "-- backend threads $bt, $args"; VVDEC_AMD_TIMES=1 VVDEC_AMD_BACKEND_THREADS=$bt LD_PRELOAD=$PWD/vvdec_amd/libvvdec_amd.so $R/vvdecapp_dropin -b $S $args -v 3 -L 4 2>&1 | grep -E "frames decoded|host ms per picture" | tail -3 | cut -c1-220; done; done
