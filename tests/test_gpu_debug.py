"""Step-replay debug mode (reference: mmat.rg -d <dir>, write_blocks mmat.rg:174-218, gen_filename :149-172,
verify.debug_factor verify.py:216-275).  cholamd_mmat -d prints the reference's Block / Cluster / Fill lines and, while it runs
the level loop one fused task at a time, the tasks' POTRF / TRSM / GEMM lines on stdout, and dumps the whole matrix after every
task under the reference's file names.  The checker below is this repo's own: it replays the logged operations with dense
numpy / scipy arithmetic on P A P^T and compares, whenever the (block, operation) changes, the block's tiles of the interval
in force with the dump of the last task -- the procedure the reference's debug_factor applies to its own runs."""
import ast
import os
import subprocess

import numpy as np
import pytest
import scipy.io
import scipy.linalg

from conftest import ROOT, case_paths

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "cholesky_amd", "bin", "cholamd_mmat")


def _region(line, which):
    lo, hi = line[f"{which}_Lo"], line[f"{which}_Hi"]
    return slice(lo[0], hi[0] + 1), slice(lo[1], hi[1] + 1)


def _dump_name(line):
    tag = {k: "%d%d" % (line[k][0], line[k][1]) for k in ("A", "B", "C") if k in line}  # gen_filename prints the colours with %d%d
    if line["op"] == "POTRF":
        return f"potrf_lvl{line['Level']}_a{tag['A']}.mtx"
    if line["op"] == "TRSM":
        return f"trsm_lvl{line['Level']}_a{tag['A']}_b{tag['B']}.mtx"
    return f"gemm_lvl{line['Level']}_a{tag['A']}_b{tag['B']}_c{tag['C']}.mtx"


def _replay(log, pmat, directory):
    blocks, clusters, ops = [], [], []
    for raw in log.splitlines():
        ln = raw.strip()
        for key in ("Block:", "Cluster:", "Fill:", "POTRF:", "TRSM:", "GEMM:"):
            if ln.startswith(key):
                d = ast.literal_eval(ln[len(key):].strip())
                if key == "Block:":
                    blocks.append(d)
                elif key == "Cluster:":
                    clusters.append(d)
                elif key != "Fill:":
                    d["op"] = key[:-1]
                    ops.append(d)
    mat = pmat.copy()
    checked = 0

    def check(last):
        nonlocal checked
        out = np.tril(scipy.io.mmread(os.path.join(directory, _dump_name(last))).toarray())
        tiles = [c for c in clusters if c["Interval"] == last["Interval"] and c["Block"] == last["Block"]]
        assert tiles, last
        for c in tiles:
            r, q = slice(c["Lo"][0], c["Hi"][0] + 1), slice(c["Lo"][1], c["Hi"][1] + 1)
            assert np.allclose(np.tril(mat)[r, q], out[r, q], rtol=1e-4, atol=1e-4), (last, c)
        checked += 1

    last = None
    for d in ops:
        # the dump of the last task of a (block, operation) run is compared BEFORE the next operation touches the matrix
        if last is not None and (last["Block"] != d["Block"] or last["op"] != d["op"]):
            check(last)
        if d["op"] == "POTRF":
            a = _region(d, "A")
            mat[a] = scipy.linalg.cholesky(mat[a], lower=True)
        elif d["op"] == "TRSM":
            a, b = _region(d, "A"), _region(d, "B")
            mat[b] = scipy.linalg.solve_triangular(mat[a], mat[b].T, lower=True).T
        else:
            a, b, c = _region(d, "A"), _region(d, "B"), _region(d, "C")
            mat[c] = mat[c] - mat[a] @ mat[b].T
            if a == b:
                mat[c] = np.tril(mat[c])
        last = d
    check(last)
    return np.tril(mat), len(blocks), len(clusters), len(ops), checked


@pytest.mark.parametrize("case", ["lapl_9x9", "lapl_25x25", "lapl_400x400"])  # lapl_400: 31 separators -- two-digit colours in gen_filename's %d%d names (mmat.rg:149-172)
def test_debug_mode_replays_step_by_step(case, tmp_path, golden):
    m, o, c, _ = case_paths(case)
    fac = tmp_path / "L.mtx"
    r = subprocess.run([BIN, "-i", m, "-s", o, "-c", c, "-d", str(tmp_path), "-m", str(fac), "--full-precision"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    g = golden(case)
    L, nblocks, nclusters, nops, checked = _replay(r.stdout, g["pmat"], str(tmp_path))
    import cholesky_amd as ca
    plan = ca.Plan(m, o, c)
    assert nblocks == len(plan.blocks) and nops == len(plan.ops()) and nclusters > 0 and checked > 0
    assert np.abs(L - g["L"]).max() <= 1e-12                      # the replay of the log reproduces the reference's L
    Lf = np.tril(scipy.io.mmread(str(fac)).toarray())
    assert np.abs(Lf - g["L"]).max() <= 1e-12                     # and so does the factor debug mode leaves behind
    # the reference's dump files exist under its names, text dumps carry its header
    names = sorted(f for f in os.listdir(tmp_path) if f.startswith(("potrf_", "trsm_", "gemm_")))
    assert any(f.endswith(".txt") for f in names) and any(f.endswith(".mtx") for f in names)
    some = [f for f in names if f.startswith("potrf_") and f.endswith(".txt")][0]
    head = open(tmp_path / some).readline()
    assert head.startswith("Level: ") and " POTRF A=(" in head
    assert "Fill: {'Level': " in r.stdout and "Partitioning (" in r.stdout and "filename: " in r.stdout
    if case == "lapl_400x400":  # colours beyond 9: "a2031" is block (20, 31) or (203, 1) -- the reference's own ambiguity, names reproduced as they are
        assert any(len(f.split("_a")[1].split("_")[0].split(".")[0]) >= 4 for f in names if f.startswith("potrf_"))
