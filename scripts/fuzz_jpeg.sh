#!/bin/bash
# Crash-safety fuzz of the JPEG decoder (opensplat_amd/csrc/gs_image.c) under AddressSanitizer + UBSan:
# every stored fixture (tests/golden/jpeg_fixtures.npz: sequential, progressive, restart markers, greyscale)
# is mutated 4000 times (1-6 random bytes, every fifth file truncated) and decoded.  Host-only.
#   bash scripts/fuzz_jpeg.sh        -> "N mutated files, M decoded", no sanitizer report
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
python3 - "$ROOT" "$T" <<'PY'
import sys, numpy as np
g = np.load(sys.argv[1] + "/tests/golden/jpeg_fixtures.npz")
for k in g.files:
    if k.endswith("_file"):
        open(sys.argv[2] + "/" + k[:-5] + ".jpg", "wb").write(g[k].tobytes())
PY
gcc -std=c11 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -I"$ROOT/include" \
    "$ROOT/scripts/fuzz_jpeg.c" "$ROOT/opensplat_amd/csrc/gs_image.c" -o "$T/fuzz"
"$T/fuzz" "$T"/*.jpg
rm -rf "$T"
