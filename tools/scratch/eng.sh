for e in 2 3 4; do
  for w in 16 32; do
    echo "== engines $e (groups on) workers $w"
    python tools/per_site_multiprocess.py --shared --engines $e --workers $w --calls 3000 2>&1 | grep -v "worker [0-9]*:\|amdgpu.ids"
    sleep 2
  done
done
