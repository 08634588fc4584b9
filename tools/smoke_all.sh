#!/bin/bash
# smoke every python tool on the GPU box: exit status, seconds, last line (what still runs, what does not)
mkdir -p gpurun_out
: > gpurun_out/r06_tools_smoke.txt
for f in tools/*.py; do
  case "$f" in tools/rocpd_*|tools/kernel_resources.py|tools/make_nav_geometry.py|tools/soak.py|tools/soak_persistent.py) continue;; esac
  s=$(date +%s)
  timeout 150 python "$f" > /tmp/smoke_out.txt 2>&1
  rc=$?
  echo "$f rc=$rc $(( $(date +%s) - s ))s :: $(tail -1 /tmp/smoke_out.txt | cut -c1-150)" >> gpurun_out/r06_tools_smoke.txt
done
cat gpurun_out/r06_tools_smoke.txt
