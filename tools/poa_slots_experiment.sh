cd "$GRAFT_REPO_ROOT"
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
B="python3 bench.py --data-cache $cache --no-stream-leg --no-shard-leg --no-cpu-baseline --bam-leg none --steps 1"
for sl in 0 1024 2048 4096; do
  if [ $sl = 0 ]; then out=$($B 2>/dev/null); else out=$(TELR_POA_SLOTS=$sl $B 2>/dev/null); fi
  echo "slots $sl: $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['te_loci']['polish_pileup']['poa']['seconds'])")"
done
