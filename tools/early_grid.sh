#!/bin/bash
# C3-size probe of the tile phase's early-stop policy: (window, tau) grid at p_work 0.1 with 2 and 3 join passes
for w in 64 32 16; do for t in 19 38 77; do
  echo "window $w tau $t"; ANNCHOR_ST_EARLY_WINDOW=$w ANNCHOR_ST_EARLY_TAU=$t python tools/stream_join_probe.py 2>&1 | grep -E "p_work 0.10 joins [23]" | cut -c1-90
done; done
