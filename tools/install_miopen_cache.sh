#!/bin/bash
# Unpacks a MIOpen user cache packed by tools/profile_train.sh on an MI355X box (gpurun_out/<tag>_miopen_cache.tgz:
# cache/gfx950100.ukdb = compiled kernels, db/*.ufdb.txt = find results) into svbrdf_estimation_amd/training/miopen_cache/,
# where train.py and the test suite pick it up (training.use_in_tree_miopen_cache).  Tracked, with a MANIFEST.json of file
# hashes and provenance: run `python tools/miopen_cache_manifest.py "<how it was produced>"` afterwards.
#   bash tools/install_miopen_cache.sh gpurun_out/r03h_miopen_cache.tgz
set -e
cd "$(dirname "$0")/.."
D=svbrdf_estimation_amd/training/miopen_cache
rm -rf $D && mkdir -p $D
tar xzf "$1" -C $D
chmod -R u+rwX,go+rX $D
rm -f $D/db/*.time
find $D -type f | xargs ls -la
