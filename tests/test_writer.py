"""PICO recon format writer (reference PICOFormatWriter/PICOFormatWriterV4.cpp).  CPU: known-answer rows of the
product writer and of the oracle's restatement.  GPU: a whole run written by the product (HIP path + C++ API mirror,
driven like main()'s event loop) must be byte-identical to the oracle's text."""
import os

import numpy as np
import pytest

from autobub3hs_amd import host, synth


@pytest.fixture(scope="module", autouse=True)
def _built():
    host.build()


HEADER = (
    "Output of AutoBub v3 - the automatic unified bubble finder code by Pitam, using OpenCV.\n"
    "run  ev  ibubimage  TotalBub4CamImg  camera  frame0  hori  vert  GenesisW  GenesisH  dZdt  dRdt  "
    "TrkFrame(10)  TrkHori(10)  TrkVert(10)  TrkBubW(10)  TrkBubH(10)  TrkBubRadius(10)  FakeValue\n"
    "%12s  %5d  %d  %d  %d  %d  %.02f  %.02f  %d  %d  %.02f  %.02f  " + "%d  " * 10 + "%.02f  " * 50 + "%d\n8\n\n\n"
)


def error_row(run, ev, cam, code):
    return f"{run}  {ev}  0  0  {cam}  {code}  0.00  0.00  0  0  0.00  0.00  " + "0  " * 10 + "0.00  " * 50 + "1  \n"


def test_header_and_error_rows(tmp_path, oracle):
    out = str(tmp_path) + "/"
    host.write_header(out, "20201005_3", 30, 2)
    txt = open(out + "abub3hs_20201005_3.txt").read()
    assert txt == HEADER == oracle.format_header()
    host.writer_probe(out, "20201005_3", 30, 17, [(-3, 0, []), (-9, 0, [])])
    txt = open(out + "abub3hs_20201005_3.txt").read()
    assert txt == HEADER + error_row("20201005_3", 17, 0, -3) + error_row("20201005_3", 17, 1, -9)


def test_bubble_rows_known_answer(tmp_path):
    out = str(tmp_path) + "/"
    g = [100, 50, 5, 7, 12.0, 1.9544, 12, 1200, 600, 102.256, 53.004]
    t1 = [99, 49, 8, 9, 30.0, 3.0902, 30, 3000, 1500, 102.5, 53.499]
    t2 = [97, 47, 11, 13, 80.0, 5.0463, 80, 8000, 4000, 102.125, 53.0]
    lone = [10, 20, 3, 3, 4.0, 1.128, 4, 44, 84, 11.0, 21.0]
    host.writer_probe(out, "R", 30, 4, [(0, 12, [[g, t1, t2]]), (0, 13, [[lone]])])
    rows = open(out + "abub3hs_R.txt").read().split("\n")
    a = rows[0].split("  ")
    # run ev ibub nTot cam frame0 x y W H dzdt drdt
    assert a[:12] == ["R", "4", "1", "2", "0", "42", "102.26", "53.00", "5", "7", "1.50", "4.24"]
    assert a[12:22] == ["43", "44", "44", "45", "46", "47", "48", "49", "50", "51"]   # 2 tracked, then 8 from the last one
    assert a[22:32] == ["102.50", "102.12"] + ["-1"] * 8
    assert a[32:42] == ["53.50", "53.00"] + ["-1"] * 8
    assert a[42:52] == ["8", "11"] + ["-1"] * 8
    assert a[52:62] == ["9", "13"] + ["-1"] * 8
    assert a[62:72] == ["3.09", "5.05"] + ["-1"] * 8
    assert a[72:] == ["1", ""]
    b = rows[1].split("  ")
    # second camera: ibubimage continues at 2; an untracked bubble prints NaN for dZdt/dRdt (0/0)
    assert b[:10] == ["R", "4", "2", "2", "1", "43", "11.00", "21.00", "3", "3"]
    assert b[10] in ("nan", "-nan") and b[11] in ("nan", "-nan")
    assert b[12:22] == [str(43 + j) for j in range(10)]
    assert b[22:72] == ["-1"] * 50


@pytest.mark.gpu
def test_run_file_byte_identical_to_oracle(tmp_path, oracle):
    W, H, F, ncams, nev, frame_offset = 640, 200, 41, 2, 6, 30
    run_number = "20200925_1"
    run = host.Run()
    models, stacks = [], {}
    for c in range(ncams):
        tr = synth.training_pairs(W, H, 8, c, F)
        mu, sg = oracle.welford(tr)
        models.append((mu, sg, len(tr)))
        run.set_model(c, mu, sg, len(tr))
    for e in range(nev):
        for c in range(ncams):
            spec = synth.random_spec(W, H, F, 700 + e, c, p_second=0.5, p_none=0.3, margin=30)
            if e == 2 and c == 1:
                fr = synth.render_event(W, H, spec, 700 + e, c)[:4]  # malformed: -9
            else:
                fr = synth.render_event(W, H, spec, 700 + e, c)
            stacks[(e, c)] = fr
            run.add_event(e, c, fr)
    out = str(tmp_path) + "/"
    host.write_header(out, run_number, frame_offset, ncams)
    expected = oracle.format_header()
    for e in range(nev):
        host.event_to_file(run, e, 100 + e, ncams, out, run_number, frame_offset)
        ans, staged = [], []
        for c in range(ncams):
            a = oracle.Analyzer(stacks[(e, c)], models[c][0], models[c][1], models[c][2])
            s, _, _ = a.any_cam_analysis()
            ans.append(a)
            staged.append(s)
        expected += oracle.format_event(ans, staged, run_number, 100 + e, frame_offset)
        for a in ans:
            a.close()
    got = open(out + f"abub3hs_{run_number}.txt").read()
    assert got == expected
    assert got.count("\n") == 6 + sum(1 for _ in got.split("\n")[6:-1])
    assert "  -9  " in got and "  -3  " in got
    run.close()
