"""abub3hs command-line driver (scope row 8f #4): argument surface and exit codes of the reference's main()
(AutoBubStart3.cpp:148-405) on CPU; on the GPU a whole run from disk (directory and zip) must produce exactly the
oracle's recon text."""
import os
import subprocess
import zipfile

import numpy as np
import pytest
from PIL import Image

from autobub3hs_amd import host, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "autobub3hs_amd", "abub3hs")


@pytest.fixture(scope="module", autouse=True)
def _built():
    host.build()


def run_cli(args, env=None, timeout=300):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([EXE] + args, capture_output=True, text=True, env=e, timeout=timeout)
    return p.returncode, p.stdout, p.stderr


def test_help_and_missing_arguments():
    rc, out, _ = run_cli(["-h"])
    assert rc == 1 and out.startswith("Usage: abub3hs")
    rc, out, _ = run_cli([])
    assert rc == 1
    rc, _, err = run_cli(["-d", "/tmp"])
    assert rc == 255 and "Insufficient required arguments" in err  # main returns -1


def test_unreadable_zip_writes_minus5_rows(tmp_path):
    rc, out, _ = run_cli(["-z", "-d", str(tmp_path), "-r", "20200101_0", "-o", str(tmp_path)], env={"ABUB_NUM_CAMS": "2"})
    assert rc == 251  # -5
    txt = open(os.path.join(tmp_path, "abub3hs_20200101_0.txt")).read().split("\n")
    assert txt[6].startswith("20200101_0  -1  0  0  0  -5  ") and txt[7].startswith("20200101_0  -1  0  0  1  -5  ")


def test_empty_run_fails_training_with_minus7(tmp_path):
    os.makedirs(os.path.join(tmp_path, "20200101_0"))
    rc, out, _ = run_cli(["-d", str(tmp_path), "-r", "20200101_0", "-o", str(tmp_path)], env={"ABUB_NUM_CAMS": "1"})
    assert rc == 249 and "Failed to train" in out  # -7


@pytest.mark.gpu
def test_cli_run_matches_oracle_text(tmp_path, oracle):
    W, H, F, ncams, nev = 320, 128, 41, 2, 5
    run_id = "20200925_1"
    rd = os.path.join(tmp_path, "data", run_id)
    stacks = {}
    for e in range(nev):
        for c in range(ncams):
            spec = synth.random_spec(W, H, F, 900 + e, c, p_none=0.2, margin=20)
            st = synth.render_event(W, H, spec, 900 + e, c)
            stacks[(e, c)] = st
            d = os.path.join(rd, str(e), "Images")
            os.makedirs(d, exist_ok=True)
            for k in range(F):
                Image.fromarray(st[k]).save(os.path.join(d, f"cam{c}_image{30 + k}.png"))
    # expected text from the oracle: train on frames 0,1 of every event, then every (event, camera)
    expected = oracle.format_header()
    models = []
    for c in range(ncams):
        tr = np.concatenate([stacks[(e, c)][:2] for e in range(nev)])
        assert all(oracle.pair_entropy16(stacks[(e, c)][1], stacks[(e, c)][0]) <= 0.0005 for e in range(nev))
        mu, sg = oracle.welford(tr)
        models.append((mu, sg, len(tr)))
    for e in range(nev):
        ans, staged = [], []
        for c in range(ncams):
            a = oracle.Analyzer(stacks[(e, c)], *models[c])
            staged.append(a.any_cam_analysis()[0])
            ans.append(a)
        expected += oracle.format_event(ans, staged, run_id, e, 30)
        for a in ans:
            a.close()
    out1 = os.path.join(tmp_path, "out_dir")
    os.makedirs(out1)
    rc, so, se = run_cli(["-d", os.path.join(tmp_path, "data"), "-r", run_id, "-o", out1, "-D", "40l-19"],
                         env={"ABUB_THREADS": "4", "ABUB_NUM_CAMS": "2"})
    assert rc == 0, (so[-2000:], se[-2000:])
    assert open(os.path.join(out1, f"abub3hs_{run_id}.txt")).read() == expected
    # same run as a zip archive
    zpath = os.path.join(tmp_path, "data", run_id + ".zip")
    with zipfile.ZipFile(zpath, "w", zipfile.ZIP_DEFLATED) as z:
        for dp, dn, fn in os.walk(rd):
            rel = os.path.relpath(dp, os.path.join(tmp_path, "data"))
            z.writestr(rel + "/", b"")
            for f in sorted(fn):
                z.write(os.path.join(dp, f), os.path.join(rel, f))
    out2 = os.path.join(tmp_path, "out_zip")
    os.makedirs(out2)
    os.rename(rd, rd + "_moved")  # make sure the zip is what gets read
    rc, so, se = run_cli(["-z", "-d", os.path.join(tmp_path, "data"), "-r", run_id, "-o", out2, "-D", "40l-19"],
                         env={"ABUB_THREADS": "2", "ABUB_NUM_CAMS": "2"})
    assert rc == 0, (so[-2000:], se[-2000:])
    assert open(os.path.join(out2, f"abub3hs_{run_id}.txt")).read() == expected
