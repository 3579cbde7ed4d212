wl=${1:-C2}; users=${2:-138493}; shift; shift
for s in "$@"; do RM_DEBUG_SPLITS=$s python3 scratch/ns.py $wl $users 3 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl splits',$s, round(d['sweep_ms'],3), round(d['frac'],4), d['tm']['sweep_blocks'])"; done
