for v in head cur head cur; do
  if [ $v = cur ]; then unset P25FE_LIB; else export P25FE_LIB=$PWD/build/abl/libp25fe_$v.so; fi
  echo "== $v 1ch"; tools/ab.sh F_$v 600 20 1 lin run 2>&1 | grep -E "k_frontend"
  echo "== $v 64ch x 60s"; tools/ab.sh G_$v 60 6 64 run 2>&1 | grep -E "k_frontend"
done
unset P25FE_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "multichannel or config4_256 or channel_batch or run_dev_matches or time_shards" 2>&1 | tail -3
