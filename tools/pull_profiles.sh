#!/bin/bash
# Build container side: copy the round's judged evidence from gpurun_out/ (scratch, merged back by gpurun) into profiles/ (tracked).
# usage: bash tools/pull_profiles.sh r04
R=${1:-r04}
cd "$(dirname "$0")/.."
for f in gpurun_out/${R}_*; do
  [ -f "$f" ] || continue
  b=profiles/$(basename "$f")
  # a measurement that was annotated in profiles/ (leading '#' lines) is not overwritten by its raw copy
  if [ -f "$b" ] && [ "$(head -c1 "$b")" = "#" ] && [ "$(head -c1 "$f")" != "#" ]; then continue; fi
  cp -v "$f" profiles/
done
