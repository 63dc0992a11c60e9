for cfg in "32 16" "64 32" "48 16" "64 24"; do set -- $cfg
  echo "window $1 tau $2"; ANNCHOR_ST_EARLY_WINDOW=$1 ANNCHOR_ST_EARLY_TAU=$2 python tools/stream_join_probe.py 2>&1 | grep "p_work 0.10 joins [23]" | cut -c1-90
done
