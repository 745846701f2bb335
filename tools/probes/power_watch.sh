#!/bin/bash
# Samples package power and shader clock (rocm-smi) while a command runs: is the chip at its power limit under the pipeline,
# under a pure MFMA stream, and how far does the shader clock fall?   usage: power_watch.sh <label> <command...>
label=$1; shift
"$@" > /tmp/pw_cmd.out 2>&1 &
pid=$!
sleep ${PW_DELAY:-4}
n=0
while kill -0 $pid 2>/dev/null && [ $n -lt ${PW_SAMPLES:-12} ]; do
  p=$(rocm-smi --showpower 2>/dev/null | grep -i -m1 "Package Power" | sed 's/.*: *//')
  c=$(rocm-smi --showclocks 2>/dev/null | grep -i -m1 "sclk" | sed 's/.*(\([0-9]*Mhz\)).*/\1/')
  echo "$label  power $p W   sclk $c"
  n=$((n+1)); sleep 1
done
wait $pid
tail -2 /tmp/pw_cmd.out | cut -c1-200
