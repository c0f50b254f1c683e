"""Builds ``libotvae_hip.so`` (the C-ABI library declared in ``include/otvae.h``) in-tree with hipcc for gfx950.

    python -m ot_vae_lightning_amd.build          # incremental
    python -m ot_vae_lightning_amd.build --force

No torch dependency: plain ``hipcc --offload-arch=gfx950 -shared -fPIC``.  hipcc cross-compiles without a GPU.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libotvae_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wno-unused-result",
         "-munsafe-fp-atomics", "-DNDEBUG"]
# per-source additions (none at present; attention.hip was tried with -fno-slp-vectorize, see its header)
FILE_FLAGS = {}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest(path, headers):
    h = hashlib.sha256()
    for p in [path] + headers:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + FILE_FLAGS.get(os.path.basename(path), [])).encode())
    return h.hexdigest()


def _compile(src, headers, force):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src + ".o")
    stamp = obj + ".sha"
    dig = _digest(path, headers)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = ["hipcc", "-x", "hip", *FLAGS, *FILE_FLAGS.get(src, []), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "otvae.h"))
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, headers, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or force or not os.path.exists(LIB):
        cmd = ["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[otvae build] {LIB} ({'rebuilt' if rebuilt else 'up to date'}; {len(srcs)} sources)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
