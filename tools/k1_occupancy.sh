#!/bin/bash
# K1 against resident waves per CU, product and the two truncated builds (make -C p25rx_amd/csrc variant TAG=abl1 DEFS=-DP25FE_ABLATE=1,
# TAG=abl6 DEFS=-DP25FE_ABLATE=6).  usage: bash tools/k1_occupancy.sh > out.txt
for pad in 42924 28924 19924 10924 5424 1924 0; do
  for tag in cur abl1 abl1s abl6; do
    lib=$PWD/build/variants/libp25fe_$tag.so; [ "$tag" == "cur" ] && lib=$PWD/p25rx_amd/libp25fe.so
    [ -f $lib ] || continue
    P25FE_K1_LDS_PAD=$pad P25FE_LIB=$lib python3 tools/k1_occupancy.py $tag 600 40 2>/dev/null
  done
done
