"""A fingerprint of the sources the bench path's kernels are built from.

profiles/*_pmc_traffic.json is collected in separate rocprofv3 --pmc passes (tools/collect_profiles.sh,
tools/summarize_profiles.py), not inside bench.py: the record carries this hash (and the commit it was taken at) so that
bench.py reports roofline.traffic only while the kernels are the ones that were measured, and null otherwise."""
import glob
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# what compress_impl / decompress_impl of a regular 3D volume run: transform, encoder, decoder, engine + headers
BENCH_PATH_SOURCES = ["xform.hip", "speck_enc.hip", "speck_dec.hip", "engine.hip"]


def bench_path_hash():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "sperr_amd", "csrc")
    files = [os.path.join(csrc, f) for f in BENCH_PATH_SOURCES] + sorted(glob.glob(os.path.join(csrc, "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def head_commit():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except Exception:   # noqa: BLE001  (the GPU box has no .git)
        return None
