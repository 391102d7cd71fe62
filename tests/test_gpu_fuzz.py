"""Randomised cases for the integer / index half of the path (SURVEY 8(a) a13, a15 and the preprocess of 8(c)): the detection
filter against the oracle's restatement of filter_detections (layers.py:264-400) over candidate counts from none to every
anchor, ties, every NMS threshold regime and output size; the uint8 preprocess against the oracle's preprocess_image over
random frame shapes (resize up, down, identity on one axis).  Bit-exact: anchor indices, scores, counts; preprocessed
floats."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from hmd_ego_pose_amd.model import Session
    from hmd_ego_pose_amd.weights import seeded_state_dict
    from oracle import decode_ref
    assert torch.cuda.is_available()
    return Session, seeded_state_dict, decode_ref


def test_filter_fuzz_against_oracle(env):
    Session, sd_of, D = env
    s = Session(sd_of(0, 0), 0, 256, 4, "fp32")
    N = s.num_anchors
    rng = np.random.Generator(np.random.PCG64(123))
    t = lambda a: torch.from_numpy(a).cuda()
    for case in range(24):
        B = int(rng.integers(1, 5))
        ncand = int(rng.choice([0, 1, 2, 7, 50, 100, 101, 300, 1000, 3000, 8000, N]))
        M = int(rng.choice([1, 10, 100, 256]))
        nms = float(rng.choice([0.0, 0.3, 0.5, 0.9]))
        thr = float(rng.choice([0.05, 0.5]))
        centers = rng.uniform(20, 236, (max(1, int(rng.integers(1, 40))), 2))
        scores = rng.uniform(0, thr, (B, N)).astype(np.float32)
        boxes = np.zeros((B, N, 4), np.float32)
        for b in range(B):
            c = centers[rng.integers(0, len(centers), N)] + rng.normal(0, 3, (N, 2))
            wh = rng.uniform(8, 60, (N, 2))
            boxes[b] = np.concatenate([c - wh / 2, c + wh / 2], 1)
            idx = rng.choice(N, min(ncand, N), replace=False)
            sc = rng.uniform(thr, 1, len(idx)).astype(np.float32)
            if case % 3 == 0 and len(idx) > 4:
                sc[: len(sc) // 2] = sc[0]                                        # equal scores: ties go to the lower anchor index
            scores[b, idx] = np.maximum(sc, np.nextafter(np.float32(thr), np.float32(1)))
        rot, tr = (rng.standard_normal((B, N, 3)).astype(np.float32) for _ in range(2))
        hand = rng.standard_normal((B, N, 63)).astype(np.float32)
        det = s.filter(t(boxes), t(scores[..., None].copy()), t(rot), t(tr), t(hand), thr, nms, M)
        torch.cuda.synchronize()
        for b in range(B):
            o = D.filter_detections(boxes[b], scores[b][:, None], rot[b], tr[b], hand[b], thr, M, nms)
            ctx = (case, b, ncand, M, nms, thr)
            assert np.array_equal(det["index"][b].cpu().numpy(), o[6]), ctx
            assert np.array_equal(det["scores"][b].cpu().numpy(), o[1]), ctx
            assert np.array_equal(det["hand"][b].cpu().numpy(), o[5]), ctx
            assert int(det["count"][b]) == int((o[6] >= 0).sum()), ctx
    s.close()


def test_best_class_filter_fuzz_against_oracle(env):
    """class_specific_filter=False (layers.py:359-362) with three classes over candidate counts from none to every anchor (the LDS sort and,
    at 512x512, the global-memory sort), ties between classes of one anchor (first argmax) and between anchors, every output row."""
    Session, sd_of, D = env
    K = 3
    t = lambda a: torch.from_numpy(a).cuda()
    for size in (256, 512):
        s = Session(sd_of(0, 0, num_classes=K), 0, size, 2, "fp32")
        N = s.num_anchors
        rng = np.random.Generator(np.random.PCG64(900 + size))
        for case in range(8):
            B = int(rng.integers(1, 3))
            ncand = int(rng.choice([0, 1, 9, 100, 2000, N]))
            M = int(rng.choice([1, 10, 100, 256]))
            nms = float(rng.choice([0.0, 0.5, 0.9]))
            if case == 0:
                ncand, M = N, 10          # every anchor a candidate: at 512x512 more than the LDS sort holds (global-memory sort)
            thr = 0.25
            cxy = rng.uniform(20, size - 20, (B, N, 2)); wh = rng.uniform(6, 70, (B, N, 2))
            boxes = np.concatenate([cxy - wh / 2, cxy + wh / 2], axis=2).astype(np.float32)
            cls = rng.uniform(0, thr, (B, N, K)).astype(np.float32)
            for b in range(B):
                idx = rng.choice(N, min(ncand, N), replace=False)
                sc = (rng.integers(17, 64, (len(idx), K)) / 64.0).astype(np.float32)          # 47 distinct scores: ties inside and between anchors
                cls[b, idx] = np.where(rng.random((len(idx), K)) < 0.6, sc, cls[b, idx])
            rot, tr = (rng.standard_normal((B, N, 3)).astype(np.float32) for _ in range(2))
            hand = rng.standard_normal((B, N, 63)).astype(np.float32)
            det = s.filter(t(boxes), t(cls), t(rot), t(tr), t(hand), thr, nms, M, class_specific_filter=False)
            torch.cuda.synchronize()
            for b in range(B):
                o = D.filter_detections(boxes[b], cls[b], rot[b], tr[b], hand[b], thr, M, nms, class_specific_filter=False)
                ctx = (size, case, b, ncand, M, nms)
                for key, want in zip(("boxes", "scores", "labels", "rotation", "translation", "hand", "index"), o):
                    assert np.array_equal(det[key][b].cpu().numpy(), want), (ctx, key)
                assert int(det["count"][b]) == int((o[6] >= 0).sum()), ctx
        s.close()


@pytest.mark.parametrize("size", [256, 512])
def test_preprocess_fuzz_against_oracle(env, size):
    Session, sd_of, D = env
    s = Session(sd_of(0, 0), 0, size, 2, "fp32")
    rng = np.random.Generator(np.random.PCG64(5 + size))
    for case in range(12):
        h, w = int(rng.integers(17, 1100)), int(rng.integers(17, 1100))
        if case % 5 == 0:
            h = size
        if case % 7 == 0:
            w = size
        img = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        want = np.stack([D.preprocess_image(f, size)[0] for f in img])
        got = s.preprocess(torch.from_numpy(img).cuda()).permute(0, 2, 3, 1).cpu().numpy()
        assert np.array_equal(got, want), (size, h, w)
    s.close()


@pytest.mark.parametrize("size", [256, 512])
def test_webrtc_frame_path_against_oracle(env, size):
    """hep_preprocess_i420_device == oracle.webrtc_frame_preprocess bit for bit over frame shapes, crops and intermediate
    sizes (the app's 256 / 512 and others), and its output feeds the forward as the NCHW view."""
    Session, sd_of, D = env
    s = Session(sd_of(0, 0), 0, size, 3, "fp32")
    rng = np.random.Generator(np.random.PCG64(17 + size))
    for case, (h, w, crop, rs) in enumerate([(480, 640, 256, 512), (720, 1280, 256, 512), (256, 256, 256, 256), (300, 402, 200, 333),
                                             (1080, 1920, 512, 512), (258, 260, 100, 777)]):
        buf = rng.integers(0, 256, (3, h * w * 3 // 2), dtype=np.uint8)
        want = np.stack([D.webrtc_frame_preprocess(f, h, w, size, crop, rs)[0] for f in buf])
        got = s.preprocess_i420(torch.from_numpy(buf).cuda(), h, w, crop, rs)
        assert got.shape == (3, 3, size, size)
        assert np.array_equal(got.permute(0, 2, 3, 1).cpu().numpy(), want), (h, w, crop, rs)
    out = s.forward(got, want_features=False)
    torch.cuda.synchronize()
    assert all(torch.isfinite(o).all() for o in out[1:])
    import ctypes
    from hmd_ego_pose_amd import _capi
    rc = _capi.lib().hep_preprocess_i420_device(s.handle, got.data_ptr(), 1, 481, 640, 256, 512, got.data_ptr(), None)
    assert rc == -1 and b"even" in _capi.lib().hep_last_error()
    s.close()


def test_webrtc_frame_path_from_several_streams_on_one_handle(env):
    """The frame path's two scratch frames belong to the handle and are used asynchronously on the CALLER's stream: calls from
    different streams (an in-flight pool) are ordered by an event recorded behind each call, so a later call never overwrites
    scratch an earlier call's kernels still read.  Eight different frame sets go through one handle on four streams back to
    back with no host synchronisation in between; every result must equal the oracle's."""
    Session, sd_of, D = env
    size, (h, w, crop, rs) = 256, (720, 1280, 256, 512)
    s = Session(sd_of(0, 0), 0, size, 3, "fp32")
    rng = np.random.Generator(np.random.PCG64(5))
    bufs = [rng.integers(0, 256, (3, h * w * 3 // 2), dtype=np.uint8) for _ in range(8)]
    dev = [torch.from_numpy(b).cuda() for b in bufs]
    streams = [torch.cuda.Stream() for _ in range(4)]
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):                                  # the first round also grows the scratch buffers
        outs = []
        for i, x in enumerate(dev):
            with torch.cuda.stream(streams[i % 4]):
                outs.append(s.preprocess_i420(x, h, w, crop, rs))
    torch.cuda.synchronize()
    for i, (b, got) in enumerate(zip(bufs, outs)):
        want = np.stack([D.webrtc_frame_preprocess(f, h, w, size, crop, rs)[0] for f in b])
        assert np.array_equal(got.permute(0, 2, 3, 1).cpu().numpy(), want), i
    s.close()
