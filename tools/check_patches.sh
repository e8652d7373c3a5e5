#!/bin/bash
# Proves that patches/*.patch (the kHIP touch-points inside a Paddle-Lite tree) apply to the reference: the touched
# files are copied from $REFERENCE into a scratch git repository under /tmp (never into this repo) and every patch is
# first checked (`git apply --check`), then applied, in order.  Exit code 0 = all apply.
set -euo pipefail
REFERENCE=${REFERENCE:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -d "$REFERENCE/lite" ] || { echo "reference tree $REFERENCE absent: cannot check"; exit 2; }
T=$(mktemp -d /tmp/khip_patches.XXXXXX)
trap 'rm -rf "$T"' EXIT
cd "$T" && git init -q .
for f in $(grep -h '^diff --git' "$ROOT"/patches/*.patch | sed 's#diff --git a/\([^ ]*\) .*#\1#' | sort -u); do
  [ -f "$REFERENCE/$f" ] || continue   # a file the patch creates (lite/kernels/hip/CMakeLists.txt ...)
  mkdir -p "$(dirname "$f")" && cp "$REFERENCE/$f" "$f"
done
git add -A >/dev/null && git -c user.email=x@y -c user.name=x commit -qm base
for p in "$ROOT"/patches/*.patch; do
  git apply --check "$p"
  git apply "$p"
  echo "applies: $(basename "$p")  ($(git diff --stat | tail -1))"
  git add -A >/dev/null && git -c user.email=x@y -c user.name=x commit -qm "$(basename "$p")"
done
echo "all $(ls "$ROOT"/patches/*.patch | wc -l) patches apply to $REFERENCE"
