run() { echo "== $1"; MCCNN_DEBUG=$1 python tools/chain_probe.py cfg2 cfg3 cfg4 2>/dev/null | grep layers | cut -c1-62 | paste -sd" "; MCCNN_DEBUG=$1 python tools/config_pipe.py cfg1 cfg2 cfg3 cfg4 2>/dev/null | cut -c1-62 | paste -sd" "; }
run ""
for o in unsorted_max_points=8192 unsorted_max_points=131072 rows_min_degree=8 rows_min_degree=32 nw_lds_pad=0 nw_lds_pad=36000 scan_bg_tiles=0 scan_bg_tiles=64 plan_min_l=8 plan_min_l=2; do run $o; done
run ""
