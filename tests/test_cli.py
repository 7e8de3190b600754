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


def test_merge_of_shard_part_files(tmp_path, oracle):
    """--merge N (no GPU involved): N part files, shard r holding the blocks of events r, r + N, .. in order, become the
    reference's one file in event order; blocks are runs of rows with the same event number, of any length."""
    run_id, N = "20200925_0", 3
    header = oracle.format_header()
    blocks = ["".join(f"{run_id}  {100 + e}  {k}  row\n" for k in range(1 + e % 3)) for e in range(8)]
    for r in range(N):
        with open(os.path.join(tmp_path, f"abub3hs_{run_id}.part{r}of{N}.txt"), "w") as f:
            f.write(header + "".join(blocks[r::N]))
    rc, so, se = run_cli(["-r", run_id, "-o", str(tmp_path), "--merge", str(N)])
    assert rc == 0, (so, se)
    assert open(os.path.join(tmp_path, f"abub3hs_{run_id}.txt")).read() == header + "".join(blocks)
    with open(os.path.join(tmp_path, f"abub3hs_{run_id}.part1of{N}.txt"), "w") as f:
        f.write("another header\n" + "".join(blocks[1::N]))
    rc, so, se = run_cli(["-r", run_id, "-o", str(tmp_path), "--merge", str(N)])
    assert rc != 0 and "different header" in se


def test_unreadable_zip_writes_minus5_rows(tmp_path):
    rc, out, _ = run_cli(["-z", "-d", str(tmp_path), "-r", "20200101_0", "-o", str(tmp_path)], env={"ABUB_NUM_CAMS": "2"})
    assert rc == 251  # -5
    txt = open(os.path.join(tmp_path, "abub3hs_20200101_0.txt")).read().split("\n")
    assert txt[6].startswith("20200101_0  -1  0  0  0  -5  ") and txt[7].startswith("20200101_0  -1  0  0  1  -5  ")


def test_empty_run_fails_training_with_minus7(tmp_path):
    os.makedirs(os.path.join(tmp_path, "20200101_0"))
    rc, out, _ = run_cli(["-d", str(tmp_path), "-r", "20200101_0", "-o", str(tmp_path)], env={"ABUB_NUM_CAMS": "1"})
    assert rc == 249 and "Failed to train" in out  # -7


def _make_run(tmp_path, oracle, W=320, H=128, F=41, ncams=2, nev=5, run_id="20200925_1"):
    """A synthetic run on disk (PNG frames) with two irregular stacks -- event 3 / cam 1 has an undecodable frame 7
    (before any bubble: status -9), event 4 / cam 0 has only 20 frames -- and the oracle's recon text for it."""
    rd = os.path.join(tmp_path, "data", run_id)
    stacks, ok = {}, {}
    for e in range(nev):
        for c in range(ncams):
            spec = synth.random_spec(W, H, F, 900 + e, c, p_none=0.2, margin=20)
            st = synth.render_event(W, H, spec, 900 + e, c)
            if (e, c) == (4, 0):
                st = st[:20]
            stacks[(e, c)] = st
            ok[(e, c)] = np.ones(len(st), np.uint8)
            d = os.path.join(rd, str(e), "Images")
            os.makedirs(d, exist_ok=True)
            for k in range(len(st)):
                path = os.path.join(d, f"cam{c}_image{30 + k}.png")
                Image.fromarray(st[k]).save(path)
                if (e, c, k) == (3, 1, 7):
                    raw = open(path, "rb").read()
                    open(path, "wb").write(raw[: len(raw) // 2])  # truncated file: GetImage == -1
                    ok[(e, c)][k] = 0
    expected = oracle.format_header()
    models = []
    for c in range(ncams):
        tr = np.concatenate([stacks[(e, c)][:2] for e in range(nev)])
        assert all(oracle.pair_entropy16(stacks[(e, c)][1], stacks[(e, c)][0]) <= 0.0005 for e in range(nev))
        mu, sg = oracle.welford(tr)
        models.append((mu, sg, len(tr)))
    blocks = []
    for e in range(nev):
        ans, staged = [], []
        for c in range(ncams):
            a = oracle.Analyzer(stacks[(e, c)], *models[c], frame_ok=ok[(e, c)])
            staged.append(a.any_cam_analysis()[0])
            ans.append(a)
        blocks.append(oracle.format_event(ans, staged, run_id, e, 30))
        for a in ans:
            a.close()
    return rd, run_id, expected, blocks


@pytest.mark.gpu
def test_cli_run_matches_oracle_text(tmp_path, oracle):
    rd, run_id, header, blocks = _make_run(tmp_path, oracle)
    expected = header + "".join(blocks)
    assert any("  -9  " in b for b in blocks)
    data = os.path.join(tmp_path, "data")

    def cli(tag, extra_args=(), env=None, zipped=False):
        out = os.path.join(tmp_path, "out_" + tag)
        os.makedirs(out)
        e = {"ABUB_THREADS": "4", "ABUB_NUM_CAMS": "2"}
        e.update(env or {})
        rc, so, se = run_cli((["-z"] if zipped else []) + ["-d", data, "-r", run_id, "-o", out, "-D", "40l-19"] + list(extra_args), env=e)
        assert rc == 0, (tag, so[-2000:], se[-2000:])
        return open(os.path.join(out, f"abub3hs_{run_id}.txt")).read(), so

    # default = the batched path (whole batches of events decoded into pinned memory, RunPipeline, ordered output)
    txt, so = cli("batched")
    assert "batched detect:" in so and txt == expected
    # several small batches (the last one padded), two worker threads sharing the GPU
    txt, so = cli("batched_small", ["--gpus", "2"], env={"ABUB_BATCH_MB": "4"})
    assert "batched detect:" in so and txt == expected
    # the reference's one-analyzer-at-a-time loop
    txt, so = cli("per_event", ["--per-event"])
    assert "batched detect" not in so and txt == expected
    # events dealt to three shards (i % 3): every shard writes its own part file in event order -- all into ONE output
    # directory, as the ranks of a multi-GPU run would --, `--merge 3` assembles the reference's single ordered file
    out = os.path.join(tmp_path, "out_sharded")
    os.makedirs(out)
    for r in (2, 0, 1):
        rc, so, se = run_cli(["-d", data, "-r", run_id, "-o", out, "-D", "40l-19", "--gpu-shard", f"{r}/3"],
                             env={"ABUB_THREADS": "4", "ABUB_NUM_CAMS": "2"})
        assert rc == 0, (r, so[-2000:], se[-2000:])
        assert open(os.path.join(out, f"abub3hs_{run_id}.part{r}of3.txt")).read() == header + "".join(blocks[r::3]), r
    assert not os.path.exists(os.path.join(out, f"abub3hs_{run_id}.txt"))
    rc, so, se = run_cli(["-r", run_id, "-o", out, "--merge", "3"])
    assert rc == 0, (so, se)
    assert open(os.path.join(out, f"abub3hs_{run_id}.txt")).read() == expected
    rc, so, se = run_cli(["-r", run_id, "-o", out, "--merge", "4"])  # a part is missing
    assert rc != 0 and "cannot read" in se
    # same run as a zip archive
    zpath = os.path.join(data, run_id + ".zip")
    with zipfile.ZipFile(zpath, "w", zipfile.ZIP_DEFLATED) as z:
        for dp, dn, fn in os.walk(rd):
            rel = os.path.relpath(dp, data)
            z.writestr(rel + "/", b"")
            for f in sorted(fn):
                z.write(os.path.join(dp, f), os.path.join(rel, f))
    os.rename(rd, rd + "_moved")  # make sure the zip is what gets read
    txt, so = cli("zip", zipped=True, env={"ABUB_THREADS": "2"})
    assert "batched detect:" in so and txt == expected
    txt, so = cli("zip_per_event", ["--per-event"], zipped=True, env={"ABUB_THREADS": "2"})
    assert txt == expected


@pytest.mark.gpu
def test_cli_debug_image_write_out(tmp_path, oracle):
    """--debug 101 (localizer + analyzer debug digits, AutoBubStart3.cpp:142): the per-event path dumps the images the
    reference dumps -- DebugPeek/ev<id>_cam<c>_{000_AvgImage,00_PreTrigFrame,0_TrigFrame,02_OvrThe6Sigma,
    3_OtsuThresholded,4_BubbleDetected}.png and $HOME/test/abub_debug/ev_<id>_[pos_|neg_|pos_filter_|neg_filter_]<frame>
    -- and they hold what the oracle computes for those stages."""
    W, H, F, run_id = 320, 128, 41, "20200925_2"
    rd = os.path.join(tmp_path, "data", run_id)
    stacks = []
    for e in range(3):
        spec = synth.EventSpec(F, t0=14 + e, bubbles=[(100 + 40 * e, 60, 40)])
        st = synth.render_event(W, H, spec, 700 + e, 0)
        stacks.append(st)
        d = os.path.join(rd, str(e), "Images")
        os.makedirs(d)
        for k in range(F):
            Image.fromarray(st[k]).save(os.path.join(d, f"cam0_image{30 + k}.png"))
    home = os.path.join(tmp_path, "home")
    os.makedirs(os.path.join(home, "test", "abub_debug"))
    os.makedirs(os.path.join(tmp_path, "DebugPeek"))
    out = os.path.join(tmp_path, "out")
    os.makedirs(out)
    e = dict(os.environ, ABUB_NUM_CAMS="1", HOME=home)
    p = subprocess.run([EXE, "-d", os.path.join(tmp_path, "data"), "-r", run_id, "-o", out, "-D", "40l-19", "--debug", "101",
                        "-e", "1"], capture_output=True, text=True, env=e, timeout=300, cwd=tmp_path)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    mu, sg = oracle.welford(np.concatenate([s[:2] for s in stacks]))
    a = oracle.Analyzer(stacks[1], mu, sg, 6)
    staged, state, bubbles = a.any_cam_analysis()
    a.close()
    assert staged == 0 and "Entropy of BkgSub" in p.stdout and "-----Start ev 1, cam 0" in p.stdout
    t = state["trig"]

    def png(path):
        return np.asarray(Image.open(path).convert("L"))

    peek = os.path.join(tmp_path, "DebugPeek", "ev1_cam0_")
    assert np.array_equal(png(peek + "000_AvgImage.png"), mu)
    assert np.array_equal(png(peek + "0_TrigFrame.png"), stacks[1][t])
    assert np.array_equal(png(peek + "00_PreTrigFrame.png"), stacks[1][t - 2])
    D = oracle.process_frame(stacks[1][t], stacks[1][t - 2], sg)
    assert np.array_equal(png(peek + "02_OvrThe6Sigma.png"), D)
    thr = max(state["loc_thres"], oracle.otsu(oracle.hist256(np.where(D > state["loc_thres"], D, 0))))
    assert np.array_equal(png(peek + "3_OtsuThresholded.png"), np.where(D > thr, 255, 0).astype(np.uint8))
    marked = png(peek + "4_BubbleDetected.png")
    d0 = bubbles[0]["desc"][0]
    assert marked[d0["y"], d0["x"]] == 255 and marked[d0["y"] + d0["h"] - 1, d0["x"] + d0["w"] - 1] == 255
    dbg = os.path.join(home, "test", "abub_debug", "ev_1_")
    for k in (1, t):  # the search dumps every frame it evaluates (two-frame offset: ref = max(k - 2, 0))
        name = f"cam0_image{30 + k}.png"
        ref = max(k - 2, 0)
        assert np.array_equal(png(dbg + name), oracle.process_frame(stacks[1][k], stacks[1][ref], sg))
        c, r, s6 = stacks[1][k].astype(int), stacks[1][ref].astype(int), 6 * sg.astype(int)
        assert np.array_equal(png(dbg + "pos_" + name), np.clip(c - r - s6, 0, 255))
        assert np.array_equal(png(dbg + "neg_" + name), np.clip(r - c - s6, 0, 255))
        assert os.path.exists(dbg + "pos_filter_" + name) and os.path.exists(dbg + "neg_filter_" + name)
    assert not os.path.exists(dbg + f"cam0_image{30 + t + 3}.png")  # nothing past the look-ahead
