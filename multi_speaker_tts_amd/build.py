"""Build libmstts_hip.so (gfx950) in-tree with hipcc.  No CPU fallback exists: if this fails the
package cannot run.

Staleness is decided by CONTENT, not by mtime: every object carries the SHA-256 of its source, the shared
headers and the compile flags (`<name>.o.hash`), the library the hash of its objects' hashes
(`libmstts_hip.so.hash`); a snapshot copied to another box (fresh mtimes) is therefore not rebuilt, an edited
source always is.  Concurrent callers (one process per GPU importing the package at once) serialise on a file lock.
"""
import fcntl
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmstts_hip.so")
SOURCES = ["gemm.hip", "gemm_bf16.hip", "skinny.hip", "cell.hip", "elementwise.hip", "lsa.hip", "decoder.hip", "audio.hip", "waveglow.hip", "ge2e.hip", "skinny_bf16.hip", "persist.hip", "persist_bwd.hip", "persist_lstm.hip", "persist_infer.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("MSTTS_EXTRA_HIPCC_FLAGS", "").split()     # (extra flags: kernel experiments only)
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "prenet_body.h"), os.path.join(CSRC, "persist_common.h"), os.path.join(CSRC, "persist_fwd_parts.h"), os.path.join(CSRC, "gemm_split.inc"), os.path.join(HERE, "..", "include", "mstts.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _want(src):
    return _sha([os.path.join(CSRC, src)] + HEADERS, " ".join(FLAGS))


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _lib_want():
    return hashlib.sha256("".join(_want(s) for s in SOURCES).encode()).hexdigest()


def needs_build():
    return not os.path.exists(LIB) or _read(LIB + ".hash") != _lib_want()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():             # another process built it while this one waited
            return LIB
        objs, procs = [], []
        for src in SOURCES:
            obj = os.path.join(CSRC, src.replace(".hip", ".o"))
            objs.append(obj)
            if not force and os.path.exists(obj) and _read(obj + ".hash") == _want(src):
                continue
            cmd = [_hipcc()] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        failed = []
        for src, obj, p in procs:
            out, _ = p.communicate()
            if verbose or p.returncode:
                sys.stderr.write(out)
            if p.returncode:
                failed.append(src)
            else:
                with open(obj + ".hash", "w") as f:
                    f.write(_want(src))
        if failed:
            raise RuntimeError("hipcc failed for: %s" % ", ".join(failed))
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        with open(LIB + ".hash", "w") as f:
            f.write(_lib_want())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
