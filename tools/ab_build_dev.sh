#!/bin/bash
# tools/ab_build_dev.sh <name> <device-only -mllvm option> [more hipcc flags]  ->  strutopy_amd/libstm_<name>.so
# Like ab_build.sh, but the -mllvm option reaches the gfx950 code generator only (a GCN scheduler name crashes the x86 host compile):
# the driver's sub-commands are printed (-###), the option is taken out of the host cc1 line, and the lines are run.
cd "$(dirname "$0")/.." || exit 1
name=$1; opt=$2; shift 2
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics -mllvm -disable-machine-licm -mllvm "$opt" "$@" \
  strutopy_amd/csrc/stm_api.hip -o strutopy_amd/libstm_$name.so -ldl -### 2>&1 | grep '^ "' > /tmp/ab_dev_$name.cmds
python3 - "$opt" /tmp/ab_dev_$name.cmds <<'PY'
import shlex, subprocess, sys
opt, path = sys.argv[1:3]
for line in open(path):
    args = shlex.split(line)
    if "-triple" in args and args[args.index("-triple") + 1].startswith("x86_64") and opt in args:
        i = args.index(opt)
        del args[i - 1:i + 1]      # "-mllvm" "<opt>"
    r = subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    if r.returncode:
        print(r.stderr[-3000:]); sys.exit(r.returncode)
PY
