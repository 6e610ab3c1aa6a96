cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-alt --no-parity --no-secondary --no-roofline --sustain-seconds 0"
for i in 1 2; do
for cfg in "-" "OVERLAP" "UNFLOW_WGRAD_GROUP=0" "UNFLOW_WGRAD_GROUP=2" "UNFLOW_WGRAD_GROUP=12" "UNFLOW_WGRAD_GROUP=100"; do
  if [ "$cfg" = "-" ]; then line=$($B 2>/dev/null | grep '^{"metric"' | tail -1)
  elif [ "$cfg" = "OVERLAP" ]; then line=$($B --overlap-adam 2>/dev/null | grep '^{"metric"' | tail -1)
  else line=$(env $cfg $B 2>/dev/null | grep '^{"metric"' | tail -1); fi
  echo "$cfg $(echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done; done
