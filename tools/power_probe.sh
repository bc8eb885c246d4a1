#!/bin/bash
# Socket power and shader clock while the headline step runs (tools/, diagnostics only): rocm-smi sampled once a second beside
# `bench.py --sustain-seconds 20`; then the same for the round-4 mode (--in-flight 2) and for a lone caller (--in-flight 1).
for nfl in 4 2 1; do
  echo "== bench.py --in-flight $nfl --sustain-seconds 20"
  python bench.py --in-flight $nfl --no-extras --no-cpu-baseline --sustain-seconds 20 > /tmp/pp_$nfl.json 2>/dev/null &
  BP=$!
  sleep 9
  for i in 1 2 3 4 5 6 7 8; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Socket Graphics Package Power" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' '
    echo
    sleep 1
  done
  wait $BP
  python -c "import json; r=json.loads(open('/tmp/pp_$nfl.json').read().strip().splitlines()[-1]); print('value', r['value'], 'sustained', r['sustained']['modexps_per_s'])"
done
