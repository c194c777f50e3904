#!/bin/bash
# What clock does the GPU run the sub-cycle at?  Starts a long bench in the background and samples the shader clock and power of
# every card of the host (sysfs hwmon; the box sees all of them, runs on one) while it runs: the busy card is the one whose clock
# leaves its idle level.  bash scripts/clock_watch.sh [bench args...]  |  bash scripts/clock_watch.sh --cmd program args...
sample() {
  for d in /sys/class/drm/card*/device; do
    f=$(cat $d/hwmon/hwmon*/freq1_input 2>/dev/null | head -1); p=$(cat $d/hwmon/hwmon*/power1_average 2>/dev/null | head -1)
    [ -z "$p" ] && p=$(cat $d/hwmon/hwmon*/power1_input 2>/dev/null | head -1)
    [ -n "$f" ] && echo -n "$(basename $(dirname $d)):$((f / 1000000))MHz/$((p / 1000000))W "
  done
  echo
}
echo "idle: $(sample)"
if [ "$1" = "--cmd" ]; then shift; "$@" > /tmp/cw_cmd.txt 2>&1 &
else python bench.py --steps ${STEPS:-1500} --warmup 3 --no-cpu-baseline --no-full-step --no-unfused "$@" > /tmp/cw_bench.json 2>/dev/null &
fi
pid=$!
for i in $(seq 1 200); do
  kill -0 $pid 2>/dev/null || break
  echo "t=$i $(sample)"
  sleep 0.25
done
wait $pid
[ -f /tmp/cw_cmd.txt ] && { cat /tmp/cw_cmd.txt; exit 0; }
tail -1 /tmp/cw_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value']/1e9, 'avg_launch_ms', d['roofline'].get('avg_launch_ms'))"
