#!/bin/bash
# Unpacks a MIOpen user cache packed by tools/build_miopen_cache.sh on an MI355X box (gpurun_out/<tag>_miopen_cache_built.tgz:
# cache/gfx950100.ukdb = compiled kernels, db/*.ufdb.txt = find results) into svbrdf_estimation_amd/training/miopen_cache/,
# where train.py and the test suite pick it up through a writable copy (training.use_in_tree_miopen_cache).  Tracked, with a MANIFEST.json of file
# hashes and provenance: run `python tools/miopen_cache_manifest.py "<how it was produced>"` afterwards.
#   bash tools/install_miopen_cache.sh gpurun_out/r05_miopen_cache_built.tgz
set -e
cd "$(dirname "$0")/.."
D=svbrdf_estimation_amd/training/miopen_cache
rm -rf $D && mkdir -p $D
tar xzf "$1" -C $D
chmod -R u+rwX,go+rX $D
rm -f $D/db/*.time
find $D -type f | xargs ls -la
