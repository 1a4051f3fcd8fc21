#!/usr/bin/env bash
# Round-3 evidence for DESIGN section 6 "Where a short-K tile's cycles go" (run through gpurun).  The probe builds of the library
# live in pix2pixhdaudiosr_amd/abl/, which .gpurunignore keeps out of the snapshot (experiment binaries do not ship): the
# script therefore BUILDS them on the GPU box first (hipcc is there; about two minutes).  Writes gpurun_out/outer/probe.log.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set: run this through gpurun}"
mkdir -p gpurun_out/outer
if [ ! -f pix2pixhdaudiosr_amd/abl/libp2phd_probe.so ] || [ ! -f pix2pixhdaudiosr_amd/abl/libp2phd_probefine.so ]; then
  bash tools/ablate_gconv.sh probe "-DP2PHD_PROBE" probefine "-DP2PHD_PROBE -DP2PHD_PROBE_FINE -DP2PHD_PROBE_DRAIN" > gpurun_out/outer/build.log 2>&1
fi
LOG=gpurun_out/outer/probe.log
: > $LOG
A="48 96 512 256 2 0"
for o in gconv_bm=0 gconv_bm=258 gconv_bm=128; do
  for lib in probe probefine; do
    P2PHD_OPTIONS=$o P2PHD_LIB=$PWD/pix2pixhdaudiosr_amd/abl/libp2phd_$lib.so python3 tools/probe_outer.py $A 2>&1 | grep -v amdgpu.ids >> $LOG
  done
  P2PHD_OPTIONS=$o python3 tools/probe_outer.py $A 2>&1 | grep -v amdgpu.ids >> $LOG
done
echo "## tools/ab_outer.py (us; tile choice forced through gconv_bm)" >> $LOG
python3 tools/ab_outer.py 2>&1 | grep -v amdgpu.ids >> $LOG
tail -n 12 $LOG
