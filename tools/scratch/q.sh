for e in 1 2; do
for c in C4 hybrid_full; do
  echo "== engines $e config $c"
  timeout -k 10 300 python tools/per_site_multiprocess.py --shared --engines $e --workers 16 --calls 2000 --config $c 2>&1 | grep aggregate
  sleep 1
done; done
