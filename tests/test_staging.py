"""Host-side staging of sweeps: the library's threaded PNG decoder (av_png_decode_gray8) against Pillow, and
the one-step-ahead frame stager on EuRoC-layout directories.  Host code only: no GPU needed."""
import os

import numpy as np
import pytest
from PIL import Image

from uav_airvision_amd import _native as N
from uav_airvision_amd.euroc import EuRoCDataset, FrameStager, decode_batch, read_image


def _images(rng, n, h=480, w=752):
    out = []
    for i in range(n):
        kind = i % 4
        if kind == 0:
            a = rng.integers(0, 256, (h, w), dtype=np.uint8)                        # noise: filter 0 / mixed
        elif kind == 1:
            a = (np.add.outer(np.arange(h), np.arange(w)) // 3 % 256).astype(np.uint8)     # ramps: Sub / Up / Paeth rows
        elif kind == 2:
            a = np.clip(128 + 60 * np.sin(np.arange(w) / 17.0)[None, :] + rng.normal(0, 3, (h, w)), 0, 255).astype(np.uint8)
        else:
            a = np.zeros((h, w), np.uint8); a[h // 3:, w // 4:] = 200
        out.append(a)
    return out


def test_png_decoder_matches_pillow_on_every_filter_type(tmp_path):
    rng = np.random.default_rng(3)
    imgs = _images(rng, 12)
    paths = []
    for i, a in enumerate(imgs):
        p = str(tmp_path / ('%d.png' % i))
        Image.fromarray(a).save(p, compress_level=[1, 6, 9][i % 3], optimize=bool(i & 1))
        paths.append(p)
    seen = set()
    for p in paths:                                                    # which row filters did the encoder choose?
        import zlib
        raw = open(p, 'rb').read()
        pos, idat = 8, b''
        while pos < len(raw):
            ln = int.from_bytes(raw[pos:pos + 4], 'big')
            if raw[pos + 4:pos + 8] == b'IDAT':
                idat += raw[pos + 8:pos + 8 + ln]
            pos += 12 + ln
        rows = zlib.decompress(idat)
        seen.update(rows[r * 753] for r in range(480))
    assert len(seen) >= 4, seen                                         # None, Sub, Up, Average / Paeth all occur
    out = np.zeros((len(paths), 480, 752), np.uint8)
    decode_batch(paths, out, threads=8)
    for a, o, p in zip(imgs, out, paths):
        assert np.array_equal(o, a) and np.array_equal(o, read_image(p))
    # a hole in the path list leaves its slot alone
    out2 = np.full((3, 480, 752), 7, np.uint8)
    decode_batch([paths[0], None, paths[2]], out2, threads=2)
    assert np.array_equal(out2[0], imgs[0]) and (out2[1] == 7).all() and np.array_equal(out2[2], imgs[2])


def _write_png_with_filters(path, img, filters, level=6, idat=1 << 30):
    """A greyscale PNG whose row r uses PNG filter type filters[r] (spec 9.2, bpp = 1): the encoder side of the decoder under test."""
    import struct, zlib
    h, w = img.shape
    a = img.astype(np.int32)
    left = np.concatenate([np.zeros((h, 1), np.int32), a[:, :-1]], 1)
    up = np.concatenate([np.zeros((1, w), np.int32), a[:-1]], 0)
    ul = np.concatenate([np.zeros((h, 1), np.int32), up[:, :-1]], 1)
    pp = left + up - ul
    pa, pb, pc = np.abs(pp - left), np.abs(pp - up), np.abs(pp - ul)
    paeth = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
    pred = [np.zeros_like(a), left, up, (left + up) >> 1, paeth]
    rows = b''.join(bytes([f]) + ((a[r] - pred[f][r]) & 0xFF).astype(np.uint8).tobytes() for r, f in enumerate(filters))
    z = zlib.compress(rows, level)

    def chunk(t, d):
        return struct.pack('>I', len(d)) + t + d + struct.pack('>I', zlib.crc32(t + d) & 0xFFFFFFFF)
    open(path, 'wb').write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 0, 0, 0, 0))
                           + b''.join(chunk(b'IDAT', z[i:i + idat]) for i in range(0, len(z), idat)) + chunk(b'IEND', b''))


@pytest.mark.parametrize('backend', ['libdeflate', 'zlib'])
def test_png_decoder_row_filter_runs_and_both_inflate_back_ends(tmp_path, backend):
    """Paeth rows are un-filtered eight at a time (SSE2 lanes), four at a time or singly depending on how many line up, and the
    deflate stream is inflated by libdeflate or by zlib (AV_PNG_ZLIB=1, read when the library is first used: a child process)."""
    import subprocess, sys, textwrap
    rng = np.random.default_rng(11)
    cases = []
    for i, (h, w) in enumerate([(480, 752), (37, 16), (41, 15), (64, 100), (9, 752)]):
        img = np.clip(rng.normal(120, 40, (h, w)) + 50 * np.sin(np.arange(w) / 9.0)[None, :], 0, 255).astype(np.uint8)
        if i == 0:            # runs of Paeth rows of every length 1..20 separated by one row of another type; the first row Paeth too
            f = [4]
            run = 1
            while len(f) < h:
                f += [4] * run + [int(rng.integers(0, 4))]
                run = run % 20 + 1
            f = f[:h]
        elif i == 4:
            f = [4] * h      # all Paeth: row 0 alone, then one group of eight
        else:
            f = [int(v) for v in rng.integers(0, 5, h)]
            f[h // 2:h // 2 + 17] = [4] * 17
        path = str(tmp_path / ('f%d.png' % i))
        _write_png_with_filters(path, img, f, level=[1, 6, 9][i % 3], idat=(1 << 30) if i % 2 else 5000)
        np.save(str(tmp_path / ('f%d.npy' % i)), img)
        assert np.array_equal(np.asarray(Image.open(path)), img)            # the writer above is a valid encoder
        cases.append((path, h, w))
    code = textwrap.dedent("""
        import sys, numpy as np
        from uav_airvision_amd.euroc import decode_batch
        for a in sys.argv[1:]:
            path, h, w = a.split(','); h = int(h); w = int(w)
            out = np.zeros((1, h, w), np.uint8)
            decode_batch([path], out)
            assert np.array_equal(out[0], np.load(path[:-4] + '.npy')), path
        print('ok')
    """)
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env.pop('AV_PNG_ZLIB', None)
    if backend == 'zlib':
        env['AV_PNG_ZLIB'] = '1'
    r = subprocess.run([sys.executable, '-c', code] + ['%s,%d,%d' % c for c in cases], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stderr[-2000:]


def test_png_decoder_other_flavours_and_errors(tmp_path):
    rng = np.random.default_rng(4)
    g = rng.integers(0, 256, (480, 752), dtype=np.uint8)
    rgb = str(tmp_path / 'rgb.png'); Image.fromarray(np.stack([g, g, g], -1)).save(rgb)
    ok = str(tmp_path / 'ok.png'); Image.fromarray(g).save(ok)
    out = np.zeros((2, 480, 752), np.uint8)
    decode_batch([rgb, ok], out)                                        # the RGB file goes through Pillow, the other natively
    assert np.array_equal(out[0], g) and np.array_equal(out[1], g)
    bad = str(tmp_path / 'bad.png')
    open(bad, 'wb').write(open(ok, 'rb').read()[:5000])
    with pytest.raises(N.AirvisionError, match='corrupt|truncated'):
        decode_batch([ok, bad], out)
    with pytest.raises(N.AirvisionError, match='cannot open'):
        decode_batch([str(tmp_path / 'missing.png'), ok], out)
    # one flipped bit inside the compressed data: the chunk CRC catches it (cv2.imread of the reference returns None for such a file,
    # dataset.py:110) -- never garbage pixels with status 0
    raw = bytearray(open(ok, 'rb').read())
    raw[len(raw) // 2] ^= 0x10
    flip = str(tmp_path / 'flip.png'); open(flip, 'wb').write(bytes(raw))
    with pytest.raises(N.AirvisionError, match='CRC|corrupt'):
        decode_batch([flip, ok], out)
    # the same image re-chunked into many small IDAT chunks (libpng writes 8 KB chunks; the stream's end marker and Adler-32 may
    # land in a chunk of their own) still decodes, and losing that last chunk is detected as a truncated stream
    import struct, zlib
    src = open(ok, 'rb').read()
    pos, idat, ihdr = 8, b'', None
    while pos < len(src):
        ln = int.from_bytes(src[pos:pos + 4], 'big')
        if src[pos + 4:pos + 8] == b'IDAT':
            idat += src[pos + 8:pos + 8 + ln]
        elif src[pos + 4:pos + 8] == b'IHDR':
            ihdr = src[pos + 8:pos + 8 + ln]
        pos += 12 + ln

    def chunk(t, d):
        return struct.pack('>I', len(d)) + t + d + struct.pack('>I', zlib.crc32(t + d) & 0xFFFFFFFF)
    body, tail = idat[:-4], idat[-4:]                                  # the Adler-32 trailer goes into a chunk of its own
    parts = [body[i:i + 8192] for i in range(0, len(body), 8192)]
    many = str(tmp_path / 'many.png')
    open(many, 'wb').write(src[:8] + chunk(b'IHDR', ihdr) + b''.join(chunk(b'IDAT', q) for q in parts) + chunk(b'IDAT', tail) + chunk(b'IEND', b''))
    decode_batch([many, ok], out)
    assert np.array_equal(out[0], g)
    cut = str(tmp_path / 'cut.png')
    open(cut, 'wb').write(src[:8] + chunk(b'IHDR', ihdr) + b''.join(chunk(b'IDAT', q) for q in parts) + chunk(b'IEND', b''))
    with pytest.raises(N.AirvisionError, match='corrupt|truncated'):
        decode_batch([cut, ok], out)
    # a 16-bit greyscale file is refused, not truncated to 8 bits
    g16 = str(tmp_path / 'g16.png'); Image.fromarray((g.astype(np.uint16) << 8)).save(g16)
    with pytest.raises(Exception, match='uint16|expected uint8'):
        decode_batch([g16, ok], out)
    small = str(tmp_path / 'small.png'); Image.fromarray(g[:100, :100]).save(small)
    with pytest.raises(Exception):
        decode_batch([small, ok], out)                                  # wrong size: Pillow's result does not fit the slot


def test_frame_stager_steps_ragged_streams_in_order(tmp_path, cfg):
    from uav_airvision_amd.euroc import write_euroc_layout
    from uav_airvision_amd.synth import SyntheticStream
    roots = []
    for i, n in enumerate((5, 3)):
        st = SyntheticStream(cfg, seed=40 + i, n_frames=n, t0=1403636580.0 + 100 * i)
        roots.append(write_euroc_layout(str(tmp_path / ('S%d' % i)), st))
    dss = [EuRoCDataset(r) for r in roots]
    ref = [list(d.stereo) for d in dss]
    stg = FrameStager(dss, 480, 752, threads=4)
    k = 0
    while True:
        nxt = stg.next()
        if nxt is None:
            break
        ts, i0, i1 = nxt
        for s in range(2):
            if k < len(ref[s]):
                assert ts[s] == ref[s][k].timestamp
                assert np.array_equal(i0[s], ref[s][k].cam0_image) and np.array_equal(i1[s], ref[s][k].cam1_image)
            else:
                assert ts[s] == -1.0 and not i0[s].any() and not i1[s].any()
        k += 1
    assert k == 5
    stg.close()
    stg = FrameStager(dss, 480, 752, max_frames=2)
    assert stg.next() is not None and stg.next() is not None and stg.next() is None
    stg.close()


def _fake_datasets(starts, n, seq='A'):
    class D(object):
        def __init__(self, s0):
            self.stereo_files = [(1.0 + 0.05 * k, '%s/cam0/%d.png' % (seq, k), '%s/cam1/%d.png' % (seq, k)) for k in range(s0, n)]
    return [D(s0) for s0 in starts]


@pytest.mark.parametrize('starts,n', [([0, 20, 40, 60, 80, 100, 120, 140], 200), ([0, 0, 7, 7, 3], 40), ([5], 30), ([0, 1, 2, 3], 4)])
def test_shared_frame_plan_decodes_every_frame_once_and_never_rewrites_a_live_entry(starts, n):
    """The offset sweep's frame sharing (run.bat:4-12, dataset.py:206-214): every distinct frame appears in exactly one `new` list,
    every stream finds its current AND its previous frame in the entry the plan names, and an entry is handed to a new frame no
    earlier than two steps after its last reader (the upload of step k + 1 is issued before step k is enqueued and waits for step k - 1)."""
    from uav_airvision_amd.euroc import SharedFramePlan
    dss = _fake_datasets(starts, n) + (_fake_datasets([0, 2], max(n // 2, 3), seq='B') if len(starts) > 2 else [])      # a second sequence shares nothing
    plan = SharedFramePlan(dss)
    files = [d.stereo_files for d in dss]
    assert plan.n_steps == max(len(f) for f in files)
    keys = [(p0, p1) for k in range(plan.n_steps) for _e, p0, p1 in plan.new[k]]
    assert len(keys) == len(set(keys)) == plan.n_frames_distinct == len({(p0, p1) for f in files for _t, p0, p1 in f})
    assert plan.n_stream_frames == sum(len(f) for f in files)
    holds, last_read = {}, {}                     # entry -> key it holds; entry -> last step that read it
    for k in range(plan.n_steps):
        for e, p0, p1 in plan.new[k]:
            assert 0 <= e < plan.n_slots
            assert last_read.get(e, -10) <= k - 2, (k, e)
            holds[e] = (p0, p1)
        for s, f in enumerate(files):
            if k >= len(f):
                assert plan.slots[k, s] == -1 and plan.ts[k, s] == -1.0
                continue
            t, p0, p1 = f[k]
            e = int(plan.slots[k, s])
            assert holds[e] == (p0, p1) and plan.ts[k, s] == t
            last_read[e] = k
            if k > 0:
                ep = int(plan.slots[k - 1, s])
                assert holds[ep] == (f[k - 1][1], f[k - 1][2]), 'the previous frame of stream %d was overwritten before step %d' % (s, k)
                last_read[ep] = k
    live = max(len(set(int(e) for e in plan.slots[k] if e >= 0) | set(int(e) for e in plan.slots[k - 1] if e >= 0 and k > 0)) for k in range(plan.n_steps))
    assert live <= plan.n_slots <= plan.n_frames_distinct


def test_shared_frame_stager_decodes_the_new_frames_of_each_step(tmp_path, cfg):
    from uav_airvision_amd.euroc import EuRoCDataset, SharedFramePlan, SharedFrameStager, read_image, write_euroc_layout
    from uav_airvision_amd.synth import SyntheticStream
    root = str(tmp_path / 'SEQ')
    write_euroc_layout(root, SyntheticStream(cfg, seed=2, n_frames=6), compress_level=1)
    dss = []
    for off in (0.0, 0.1):
        d = EuRoCDataset(root); d.set_starttime(off); dss.append(d)
    plan = SharedFramePlan(dss, max_frames=5)
    assert plan.n_steps == 5 and [len(x) for x in plan.new] == [2, 2, 1, 1, 0] and plan.n_stream_frames == 9
    stg = SharedFrameStager(plan, 480, 752, threads=2)
    for k in range(plan.n_steps):
        ent, i0, i1 = stg.get(k)
        assert ent.tolist() == [e for e, _a, _b in plan.new[k]] and i0.shape == (len(ent), 480, 752) == i1.shape
        for j, (_e, p0, p1) in enumerate(plan.new[k]):
            assert np.array_equal(i0[j], read_image(p0)) and np.array_equal(i1[j], read_image(p1))
    stg.close()
