#!/bin/bash
# One GPU session step with its post-mortem: runs the command from the repo root (under `timeout -k 10`), logs to
# gpurun_out/<name>.txt, and -- if the runtime left a GPU core dump (a "Memory access fault by GPU" does:
# "GPU core dump created: gpucore.N") -- opens it with rocgdb in batch mode and keeps the text next to the log, so
# that the faulting agent, wave, PC and kernel are on record without running anything again (round 3 lost gpucore.919
# by never looking at it).  The dump itself stays on the box (it can exceed what gpurun merges back).
#   usage: tools/gpu_session.sh <name> <seconds> <command ...>
set -u
name=$1; secs=$2; shift 2
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 2
mkdir -p gpurun_out
log=gpurun_out/$name.txt
timeout -k 10 "$secs" "$@" > "$log" 2>&1
rc=$?
echo "[gpu_session] exit code $rc" >> "$log"
shopt -s nullglob
for core in gpucore.* /tmp/gpucore.* "$HOME"/gpucore.*; do
    out=gpurun_out/$name.$(basename "$core").rocgdb.txt
    {
        echo "# $core ($(stat -c %s "$core") bytes), left by: $*"
        timeout -k 5 180 /opt/rocm/bin/rocgdb -batch -ex "core-file $core" -ex "info agents" -ex "info queues" \
            -ex "info dispatches" -ex "info threads" -ex "thread apply all bt 4" -ex "info sharedlibrary" /usr/bin/python3 2>&1 | tail -n 400
    } > "$out"
    echo "[gpu_session] GPU core dump $core -> $out" >> "$log"
    mv "$core" "$core.seen" 2>/dev/null
done
tail -n 40 "$log"
exit $rc
