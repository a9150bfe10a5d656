for cfg in "2 120 1" "2 60 1" "2 120 0" "1 120 0" "1 60 0" "1 200 0"; do
  set -- $cfg
  for w in 16 32; do
    echo "== engines $1, linger $2 us, groups $3, workers $w"
    HELLO_SHARED_LINGER_US=$2 HELLO_SITE_GROUPS=$3 python tools/per_site_multiprocess.py --shared --engines $1 --workers $w --calls 3000 2>&1 | grep -v "worker [0-9]*:\|amdgpu.ids"
    sleep 2
  done
done
