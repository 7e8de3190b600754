"""Frame ingestion (scope row 8f #2): PNG/BMP grey decode and the Raw / Zip parsers, checked on CPU against
Pillow decodes and python's zipfile; plus (GPU) a run read from disk analysed exactly like the same run fed
from memory."""
import io
import os
import struct
import zipfile

import numpy as np
import pytest
from PIL import Image

from autobub3hs_amd import host, synth

rng = np.random.RandomState(42)


@pytest.fixture(scope="module", autouse=True)
def _built():
    host.build()


def png_bytes(img, mode="L", **kw):
    b = io.BytesIO()
    im = Image.fromarray(img, mode="L")
    if mode == "P":
        im = im.convert("P")  # grey palette
    elif mode == "RGB":
        im = im.convert("RGB")
    elif mode == "I;16":
        im = Image.fromarray(img.astype(np.uint16) * 257)
    elif mode == "1":
        im = Image.fromarray(img > 127)
    im.save(b, format="PNG", **kw)
    return b.getvalue()


@pytest.mark.parametrize("shape", [(37, 53), (64, 1280), (5, 1)])
def test_png_decode_matches_pillow(shape):
    img = rng.randint(0, 256, shape).astype(np.uint8)
    img[: shape[0] // 2] = np.sort(img[: shape[0] // 2], axis=1)  # smoother rows: exercises filter types
    for mode in ("L", "P", "RGB", "I;16", "1"):
        for level in (0, 6, 9):
            data = png_bytes(img, mode, compress_level=level)
            got = host.imdecode(data)
            if mode == "1":
                exp = ((img > 127) * 255).astype(np.uint8)
            else:
                exp = img
            assert got is not None and np.array_equal(got, exp), (mode, level)
    assert host.imdecode(b"not an image") is None
    assert host.imdecode(png_bytes(img)[:50]) is None  # truncated


def test_bmp_decode_matches_pillow():
    img = rng.randint(0, 256, (33, 47)).astype(np.uint8)
    for mode in ("L", "P", "RGB", "1"):
        b = io.BytesIO()
        im = Image.fromarray(img)
        if mode == "1":
            im = Image.fromarray(img > 127)
        elif mode != "L":
            im = im.convert(mode)
        im.save(b, format="BMP")
        got = host.imdecode(b.getvalue())
        exp = ((img > 127) * 255).astype(np.uint8) if mode == "1" else img
        assert np.array_equal(got, exp), mode


REF = "/root/reference/cam_masks/40l-19"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference sample images are not on this machine")
def test_reference_sample_frames_and_masks_decode_like_pillow():
    for name in sorted(os.listdir(REF)):
        if not name.endswith((".png", ".bmp")):
            continue
        data = open(os.path.join(REF, name), "rb").read()
        got = host.imdecode(data, cap=1 << 23)
        exp = np.array(Image.open(os.path.join(REF, name)).convert("L"))
        assert got is not None and got.shape == exp.shape, name
        assert np.array_equal(got, exp), name


def make_run_dir(root, W=96, H=64, F=12, nev=3, ncams=2, fmt="png"):
    frames = {}
    run_id = "20200925_1"
    rd = os.path.join(root, run_id)
    for e in range(nev):
        for c in range(ncams):
            spec = synth.random_spec(W, H, F, 300 + e, c, margin=10)
            st = synth.render_event(W, H, spec, 300 + e, c)
            d = os.path.join(rd, str(e), "Images")
            os.makedirs(d, exist_ok=True)
            for k in range(F):
                name = f"cam{c}_image{30 + k}.{fmt}"
                Image.fromarray(st[k]).save(os.path.join(d, name))
                frames[(e, c, name)] = st[k]
    with open(os.path.join(rd, run_id + ".txt"), "w") as f:
        for e in range(nev):
            f.write(f"{run_id} {e} a b c d e f g h i\n")
    os.makedirs(os.path.join(rd, "9", "Images"))  # an event directory the run file does not list
    return rd, frames


def zip_run(rd, path, compress):
    root = os.path.dirname(rd)
    with zipfile.ZipFile(path, "w", compression=compress, allowZip64=True) as z:
        for dp, dn, fn in os.walk(rd):
            rel = os.path.relpath(dp, root)
            z.writestr(rel + "/", b"")
            for f in sorted(fn):
                z.write(os.path.join(dp, f), os.path.join(rel, f))


@pytest.mark.parametrize("compress", [zipfile.ZIP_STORED, zipfile.ZIP_DEFLATED])
def test_raw_and_zip_parsers_agree(tmp_path, compress):
    rd, frames = make_run_dir(str(tmp_path))
    raw = host.Run("raw", rd + "/", "Images")
    zpath = os.path.join(str(tmp_path), "run.zip")
    zip_run(rd, zpath, compress)
    zp = host.Run("zip", zpath, "Images")
    assert raw.events() == ["0", "1", "2", "9"]
    assert zp.events() == ["0", "1", "2", "9"]
    for e in range(3):
        for c in range(2):
            fr = raw.frames(e, c)
            assert fr == sorted(n for (ee, cc, n) in frames if ee == e and cc == c)  # lexicographic
            assert zp.frames(e, c) == fr
            for name in fr[:3]:
                rc1, im1 = raw.image(e, name)
                rc2, im2 = zp.image(e, name)
                assert rc1 == 1 and rc2 == 1
                assert np.array_equal(im1, frames[(e, c, name)]) and np.array_equal(im2, frames[(e, c, name)])
    assert raw.image(0, "cam0_image999.png")[0] == 0   # RawParser: 0 = empty (never -1)
    assert zp.image(0, "cam0_image999.png")[0] == -1   # ZipParser: -1 = missing
    with pytest.raises(RuntimeError):
        host.Run("zip", os.path.join(str(tmp_path), "nope.zip"), "Images")
    raw.close()
    zp.close()


def test_zip64_archive(tmp_path):
    # force zip64 records for a small archive: extra fields + zip64 EOCD must be parsed
    rd, frames = make_run_dir(str(tmp_path), nev=1, ncams=1, F=4)
    zpath = os.path.join(str(tmp_path), "z64.zip")
    root = os.path.dirname(rd)
    with zipfile.ZipFile(zpath, "w", zipfile.ZIP_DEFLATED, allowZip64=True) as z:
        for dp, dn, fn in os.walk(rd):
            rel = os.path.relpath(dp, root)
            z.writestr(rel + "/", b"")
            for f in sorted(fn):
                with z.open(zipfile.ZipInfo(os.path.join(rel, f)), "w", force_zip64=True) as dst:
                    dst.write(open(os.path.join(dp, f), "rb").read())
    zp = host.Run("zip", zpath, "Images")
    fr = zp.frames(0, 0)
    assert len(fr) == 4
    rc, im = zp.image(0, fr[0])
    assert rc == 1 and np.array_equal(im, frames[(0, 0, fr[0])])
    zp.close()


@pytest.mark.gpu
def test_run_from_disk_equals_run_from_memory(tmp_path, oracle):
    W, H, F = 320, 128, 20
    rd, frames = make_run_dir(str(tmp_path), W=W, H=H, F=F, nev=4, ncams=1)
    zpath = os.path.join(str(tmp_path), "run.zip")
    zip_run(rd, zpath, zipfile.ZIP_DEFLATED)
    results = []
    for kind, src in (("raw", rd + "/"), ("zip", zpath)):
        run = host.Run(kind, src, "Images")
        st, tss, mu, sg = run.train(0, shape=(H, W))
        assert st == 0 and tss == 2 * 4  # events 0..3 train; the frame-less event 9 is skipped (Trainer.cpp:271-274)
        out = [run.analyze(e, 0)[:3] for e in range(4)]
        results.append((tss, mu, sg, out))
        run.close()
    assert results[0][0] == results[1][0]
    assert np.array_equal(results[0][1], results[1][1]) and np.array_equal(results[0][2], results[1][2])
    assert repr(results[0][3]) == repr(results[1][3])  # repr: NaN-safe (untracked bubbles have dZdt = NaN)
    tss, mu, sg, out = results[0]
    for e in range(4):
        names = sorted(n for (ee, cc, n) in frames if ee == e)
        st = np.stack([frames[(e, 0, n)] for n in names])
        a = oracle.Analyzer(st, mu, sg, tss)
        ref = a.any_cam_analysis()
        a.close()
        assert out[e][0] == ref[0] and out[e][1] == ref[1]


@pytest.mark.gpu
def test_batched_run_survives_a_failing_batch(tmp_path, monkeypatch):
    """RunBatched decodes batch b + 1 on a look-ahead thread while batch b is on the GPU.  When batch b fails, the
    worker must join that thread before anything it writes to goes out of scope: the call returns an error, the process
    stays healthy, and the same run then goes through and gives the same text as an undisturbed one."""
    W, H, F = 320, 128, 20
    rd, _ = make_run_dir(str(tmp_path), W=W, H=H, F=F, nev=8, ncams=1)

    def go(outdir):
        os.makedirs(outdir, exist_ok=True)
        run = host.Run("raw", rd + "/", "Images")
        try:
            assert run.train(0, shape=(H, W))[0] == 0
            return run.run_batched(1, outdir + "/", "r", 30, nthreads=4, decode_threads=4, batch_mb=1)
        finally:
            run.close()

    st = go(os.path.join(str(tmp_path), "ok"))
    assert st["batches"] >= 3
    monkeypatch.setenv("ABUB_TEST_FAIL_BATCH", "0")
    with pytest.raises(RuntimeError, match="injected failure"):
        go(os.path.join(str(tmp_path), "fail"))
    monkeypatch.delenv("ABUB_TEST_FAIL_BATCH")
    go(os.path.join(str(tmp_path), "again"))
    a = open(os.path.join(str(tmp_path), "ok", "abub3hs_r.txt")).read()
    b = open(os.path.join(str(tmp_path), "again", "abub3hs_r.txt")).read()
    assert a == b and len(a.splitlines()) >= 8


@pytest.mark.gpu
def test_batched_run_with_frames_decoded_on_the_gpu(tmp_path, monkeypatch):
    """The batched run decodes PNG frames on the GPU (abub_png_decode_dev) when the parser hands out the files; host threads
    then only read them.  Same text as with host decode, from a directory, a stored and a deflated archive; a 16-bit PNG, a
    truncated file, an empty file and a frame of another size among the frames take the host decoder's answer."""
    W, H, F = 320, 128, 20
    rd, frames = make_run_dir(str(tmp_path), W=W, H=H, F=F, nev=6, ncams=2)
    d3 = os.path.join(rd, "3", "Images")
    Image.fromarray((frames[(3, 0, "cam0_image37.png")].astype(np.uint16) << 8)).save(os.path.join(d3, "cam0_image37.png"))  # 16-bit grey
    data = open(os.path.join(d3, "cam1_image40.png"), "rb").read()
    open(os.path.join(d3, "cam1_image40.png"), "wb").write(data[:len(data) // 2])  # truncated
    open(os.path.join(rd, "4", "Images", "cam0_image33.png"), "wb").close()  # empty
    Image.fromarray(np.zeros((H, W + 4), np.uint8)).save(os.path.join(rd, "4", "Images", "cam1_image35.png"))  # another size
    Image.fromarray(frames[(5, 0, "cam0_image38.png")]).convert("P").save(os.path.join(rd, "5", "Images", "cam0_image38.png"))  # palette
    zs, zd = os.path.join(str(tmp_path), "stored.zip"), os.path.join(str(tmp_path), "deflated.zip")
    zip_run(rd, zs, zipfile.ZIP_STORED)
    zip_run(rd, zd, zipfile.ZIP_DEFLATED)

    def go(kind, src, tag, gpu):
        monkeypatch.setenv("ABUB_GPU_DECODE", "1" if gpu else "0")
        outdir = os.path.join(str(tmp_path), tag)
        os.makedirs(outdir, exist_ok=True)
        run = host.Run(kind, src, "Images")
        try:
            for c in range(2):
                assert run.train(c, shape=(H, W))[0] == 0
            st = run.run_batched(2, outdir + "/", "r", 30, nthreads=4, decode_threads=4, batch_mb=2)
        finally:
            run.close()
        return st, open(os.path.join(outdir, "abub3hs_r.txt")).read()

    st0, ref = go("raw", rd + "/", "host", False)
    assert st0["frames_gpu_decoded"] == 0 and st0["frames_failed"] == 3
    for kind, src, tag in (("raw", rd + "/", "gpu_raw"), ("zip", zs, "gpu_stored"), ("zip", zd, "gpu_deflated")):
        st, text = go(kind, src, tag, True)
        assert text == ref, tag
        assert st["frames_failed"] == 3 and st["frames_host_decoded"] == 1, (tag, st)  # the 16-bit frame
        assert st["frames_gpu_decoded"] == 6 * 2 * F - 4, (tag, st)
    assert len(ref.splitlines()) >= 12
    # batches shared between the GPU (the first two events of each) and the host threads (the third)
    monkeypatch.setenv("ABUB_GPU_DECODE_EVENTS", "2")
    monkeypatch.setenv("ABUB_HOST_DECODE_EVENTS", "1")
    outdir = os.path.join(str(tmp_path), "hybrid")
    os.makedirs(outdir)
    monkeypatch.setenv("ABUB_GPU_DECODE", "1")
    run = host.Run("zip", zs, "Images")
    try:
        for c in range(2):
            assert run.train(c, shape=(H, W))[0] == 0
        st = run.run_batched(2, outdir + "/", "r", 30, nthreads=4, decode_threads=4, batch_mb=64)
    finally:
        run.close()
    assert open(os.path.join(outdir, "abub3hs_r.txt")).read() == ref
    assert st["events_per_batch"] == 3 and st["batches"] == 3, st  # (7 event directories: 0..5 and the frame-less 9)
    assert st["frames_failed"] == 3 and st["frames_gpu_decoded"] + st["frames_host_decoded"] == 6 * 2 * F - 3, st
    assert st["frames_host_decoded"] >= 2 * 2 * F - 3, st


def test_png_walk_for_the_gpu_decoder():
    """What RunBatched's reading threads find out about a file (host/pngwalk.hpp): the IDAT chunks and the palette table of
    8-bit grey / palette PNGs of the run's geometry; everything else is left to the host decoder.  The Python helper of the
    GPU tests (hip.png_parse) must say the same."""
    import io

    from autobub3hs_amd import hip

    rs = np.random.RandomState(3)
    W, H = 64, 20
    img = rs.randint(0, 256, (H, W)).astype(np.uint8)

    def png(im, **kw):
        b = io.BytesIO()
        im.save(b, format="PNG", **kw)
        return b.getvalue()

    grey = png(Image.fromarray(img))
    segs, lut = host.png_walk(grey, W, H)
    assert lut is None and len(segs) >= 1 and (segs, lut) == hip.png_parse(grey, W, H)
    o, n = segs[0]
    assert grey[o - 4:o] == b"IDAT" and int.from_bytes(grey[o - 8:o - 4], "big") == n
    pal = png(Image.fromarray(img).convert("P"))
    segs, lut = host.png_walk(pal, W, H)
    assert lut is not None and (segs, lut) == hip.png_parse(pal, W, H)
    sample = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_40l19_cam1_image30.png"), "rb").read()
    segs, lut = host.png_walk(sample, 1680, 1050)
    assert len(segs) == 15 and lut == bytes(range(256)) and (segs, lut) == hip.png_parse(sample, 1680, 1050)
    assert host.png_walk(sample, 1680, 1048) is None  # another geometry
    for other in (png(Image.fromarray(img.astype(np.uint16) << 8)), png(Image.fromarray(np.stack([img] * 3, -1))), b"BM" + bytes(60),
                  grey[:40], b"", png(Image.fromarray(img), compress_level=0)[:-20]):
        assert (host.png_walk(other, W, H) is None) == (hip.png_parse(other, W, H) is None)
    assert host.png_walk(png(Image.fromarray(img.astype(np.uint16) << 8)), W, H) is None
    assert host.png_walk(grey[:40], W, H) is None


@pytest.mark.parametrize("ext", ["png", "bmp"])
def test_imwrite_round_trip(tmp_path, ext):
    """Debug image write-out (AnalyzerUnit.cpp:237, L3Localizer.cpp:236-257): what cvlite writes, Pillow and cvlite's
    own decoder read back unchanged; a missing directory gives False like cv::imwrite."""
    from PIL import Image

    rs = np.random.RandomState(4)
    for shape in [(37, 53), (1, 1), (64, 100)]:
        img = rs.randint(0, 256, shape).astype(np.uint8)
        path = os.path.join(tmp_path, f"dbg_{shape[0]}.{ext}")
        assert host.imwrite(path, img)
        assert np.array_equal(np.asarray(Image.open(path).convert("L")), img)
        assert np.array_equal(host.imdecode(open(path, "rb").read()), img)
    assert not host.imwrite(os.path.join(tmp_path, "missing_dir", "x." + ext), img)
