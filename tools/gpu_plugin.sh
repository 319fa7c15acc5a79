#!/bin/bash
# Run a command on the MI355X box WITH the reference package `kge` present (never committed): oracle/make_ref.py
# (the recipe __graft_entry__.build() also runs) places it in the git-ignored oracle/_ref/libkge/, where
# oracle/ref_harness.py finds it; the snapshot carries it to the box.  Usage (build container, repo root):
#   bash tools/gpu_plugin.sh [--timeout S] -- '<command on the GPU box>'
set -u
cd "$(dirname "$0")/.."
python oracle/make_ref.py
exec /usr/local/graft/bin/gpurun "$@"
