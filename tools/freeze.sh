#!/bin/bash
# Freeze the committed tree + the libraries built from it into .frozen/ (git-ignored, travels with gpurun snapshots).
# A GPU call that runs `cd .frozen && ...` then tests exactly the commit named in .frozen/FROZEN_HEAD, whatever state
# the working tree around it is in (the device pool opens and closes outside the build's control; the call that gets
# through must not depend on the edit in progress).  Run it right after `make` + `git commit` on a clean tree.
set -e
cd "$(dirname "$0")/.."
if [ -z "$FORCE" ] && [ -n "$(git status --porcelain --untracked-files=no)" ]; then echo "freeze: working tree not clean" >&2; exit 1; fi
rm -rf .frozen.new && mkdir .frozen.new
git archive HEAD | tar -x -C .frozen.new
for d in nsparse_amd/lib nsparse_amd/lib_asan nsparse_amd/lib_exp; do
  [ -d "$d" ] || continue
  mkdir -p .frozen.new/$d
  # objects stay behind: only what is loaded or executed travels
  find "$d" -maxdepth 1 -type f -exec cp -p {} .frozen.new/$d/ \;
done
# variant libraries (bisect points lib_<commit>, A/B builds): kept outside the snapshot, ${NSPARSE_VARIANTS:-/tmp/nsp_variants}/lib_*
for d in ${NSPARSE_VARIANTS:-/tmp/nsp_variants}/lib_*/; do
  [ -d "$d" ] || continue
  mkdir -p .frozen.new/nsparse_amd/$(basename $d)
  find "$d" -maxdepth 1 -type f -name "*.so" -exec cp -p {} .frozen.new/nsparse_amd/$(basename $d)/ \;
done
cp -p oracle/*.so .frozen.new/oracle/ 2>/dev/null || true
git rev-parse --short HEAD > .frozen.new/FROZEN_HEAD
rm -rf .frozen && mv .frozen.new .frozen
du -sh .frozen | cut -f1
