for v in "" "HSA_ENABLE_INTERRUPT=0"; do
  for w in 16 32; do
    echo "== env '$v' workers $w"
    env $v python tools/per_site_multiprocess.py --shared --workers $w --calls 3000 2>&1 | grep -v "worker [0-9]*:\|amdgpu.ids"
    sleep 2
  done
done
echo "== one-site latency (direct engine), default vs HSA_ENABLE_INTERRUPT=0"
python tools/one_site_profile.py 2>&1 | tail -4
HSA_ENABLE_INTERRUPT=0 python tools/one_site_profile.py 2>&1 | tail -4
