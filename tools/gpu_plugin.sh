#!/bin/bash
# Run a command on the MI355X box WITH the reference package `kge` present (never committed):
# copy /root/reference/kge into the git-ignored oracle/_ref/libkge/ (oracle/ref_harness.py finds it
# there), call gpurun, remove the copy again.  Usage (build container, repo root):
#   bash tools/gpu_plugin.sh [--timeout S] -- '<command on the GPU box>'
set -u
cd "$(dirname "$0")/.."
mkdir -p oracle/_ref/libkge
cp -r /root/reference/kge oracle/_ref/libkge/kge
find oracle/_ref/libkge -name __pycache__ -type d -exec rm -rf {} + 2>/dev/null
/usr/local/graft/bin/gpurun "$@"
rc=$?
rm -rf oracle/_ref/libkge
exit $rc
