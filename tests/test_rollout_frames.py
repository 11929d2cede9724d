"""Frame capture around the path: the batched form of rollout.Controller (rollout.py:30-114) and its
frame dumps.  The PNG / npy writers are host code (CPU tests); the rollout itself runs on the GPU against the oracle."""
import os
import struct
import zlib

import numpy as np
import pytest

import golden_util as G
from oracle import pyoracle
from sequential_social_dilemma_games_amd import constants as K


def _decode_png(data):
    """Minimal reader for what save_frames_png writes: 8-bit RGB, filter 0, no interlace."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(tag + body) & 0xFFFFFFFF
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype, comp, filt, lace = hdr
    assert (depth, ctype, comp, filt, lace) == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert not raw[:, 0].any()
    return raw[:, 1:].reshape(h, w, 3)


def test_png_and_npy_dumps_round_trip(tmp_path):
    from sequential_social_dilemma_games_amd import rollout as R
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, size=(4, 16, 38, 3), dtype=np.uint8)
    names = R.save_frames_png(frames, str(tmp_path / "png"))
    assert [os.path.basename(n) for n in names] == ["frame%06d.png" % i for i in range(4)]      # rollout.py:75
    for i, n in enumerate(names):
        np.testing.assert_array_equal(_decode_png(open(n, "rb").read()), frames[i])
    big = R.save_frames_png(frames[:1], str(tmp_path / "x3"), scale=3)
    np.testing.assert_array_equal(_decode_png(open(big[0], "rb").read()), np.repeat(np.repeat(frames[0], 3, 0), 3, 1))
    full = rng.integers(0, 256, size=(4, 3, 5, 7, 3), dtype=np.uint8)
    for e, n in enumerate(R.save_frames_npy(full, str(tmp_path / "npy"))):
        np.testing.assert_array_equal(np.load(n), full[:, e])
    with pytest.raises(ValueError):
        R.save_frames_png(frames[0], str(tmp_path / "bad"))


def test_make_video_without_opencv_writes_frames(tmp_path):
    from sequential_social_dilemma_games_amd import rollout as R
    try:
        import cv2  # noqa: F401
        pytest.skip("OpenCV present: the mp4 branch is the reference's own writer")
    except ImportError:
        pass
    frames = np.zeros((2, 4, 4, 3), np.uint8)
    out = R.make_video(frames, str(tmp_path), video_name="t")
    assert sorted(os.listdir(out)) == ["frame000000.png", "frame000001.png"]


def test_controller_rejects_unknown_env():
    from sequential_social_dilemma_games_amd import rollout as R
    with pytest.raises(ValueError):
        R.Controller(env_name="watershed")


@pytest.mark.gpu
@pytest.mark.parametrize("env_name", ["harvest", "cleanup"])
def test_controller_rollout_matches_oracle(env_name, tmp_path):
    """Rewards, observations and every env's full frame of a 40-step random rollout equal the oracle's."""
    from sequential_social_dilemma_games_amd import rollout as R
    gid, amap = (K.GAME_HARVEST, K.HARVEST_MAP) if env_name == "harvest" else (K.GAME_CLEANUP, K.CLEANUP_MAP)
    E, N, T = 6, 5, 40
    c = R.Controller(env_name=env_name, num_envs=E, num_agents=N, seed=21)
    ora = pyoracle.Oracle(gid, amap, E, N, G.default_lut(), seed=21)
    ora.reset()
    rewards, observations, full_obs = c.rollout(horizon=T, save_path=str(tmp_path / "frames"))
    rewards, observations, full_obs = rewards.cpu().numpy(), observations.cpu().numpy(), full_obs.cpu().numpy()
    assert full_obs.shape == (T, E, c.engine.H, c.engine.W, 3)
    lut = G.default_lut()
    for t in range(T):
        _, obs, rew, _ = ora.step_random()
        np.testing.assert_array_equal(observations[t], obs)
        np.testing.assert_array_equal(rewards[t], rew)
        st = ora.get_state()
        for e in range(E):
            grid = st["world"][e].copy()
            for i in range(N):
                grid[st["pos"][e, i, 0], st["pos"][e, i, 1]] = ord("12345"[i])
            grid = np.where(st["beam"][e] != 0, st["beam"][e], grid)
            np.testing.assert_array_equal(full_obs[t, e], lut[grid.astype(np.int64)])
    saved = sorted(os.listdir(str(tmp_path / "frames")))
    assert saved == ["frame%06d.png" % i for i in range(T)]
    np.testing.assert_array_equal(_decode_png(open(str(tmp_path / "frames" / saved[-1]), "rb").read()), full_obs[-1, 0])
    # a sub-range of envs gives the same frames
    c2 = R.Controller(env_name=env_name, num_envs=E, num_agents=N, seed=21)
    _, _, sub = c2.rollout(horizon=5, envs=(2, 3))
    np.testing.assert_array_equal(sub.cpu().numpy(), full_obs[:5, 2:5])
