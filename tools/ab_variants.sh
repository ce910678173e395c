#!/bin/bash
# A/B several compile-time variants of ONE kernel source in ONE gpurun call (a call costs ~20 s of GPU budget before the command starts).
# Builds gshell_amd/lib/variants/<name>.so here (hipcc cross-compiles, no GPU), then runs `tool` once per variant + once on the tree's library.
# usage (CPU box):  tools/ab_variants.sh <source.hip> <tool.py> <grep pattern> name1:"-DFOO=1" name2:"-DFOO=2 -DBAR=0" ...
# e.g.              tools/ab_variants.sh envshade.hip tools/shade_time.py "shade_fwd|iteration" w2:-DTRACE_WAVES=2 w4:-DTRACE_WAVES=4
set -e
cd "$(dirname "$0")/.."
src="$1"; tool="$2"; pat="$3"; shift 3
names=""
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  tools/build_variant.sh "$name" "$src" $flags > /dev/null
  names="$names $name"
done
cmd="echo == tree; python $tool 2>&1 | grep -E \"$pat\"; for v in$names; do echo == \$v; GSHELL_HIP_LIB=gshell_amd/lib/variants/\$v.so python $tool 2>&1 | grep -E \"$pat\"; done"
/usr/local/graft/bin/gpurun --timeout 600 -- "timeout 580 bash -c '$cmd'" 2>&1 | tail -$((4 + 3 * ($# + 1)))
rm -rf gshell_amd/lib/variants
