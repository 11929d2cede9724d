"""Random-policy rollouts with full-frame capture: the batched form of the reference's `rollout.Controller`
(rollout.py:30-114), which steps ONE env with uniform random actions, takes `map_to_colors()` of the whole grid after
every step (rollout.py:77-78) and hands the frames to a video writer.

Here E envs step at once (ssd_step_random) and ssd_render_frames renders all their grids in one launch per step into
a device tensor [horizon, E, H, W, 3]; nothing crosses PCIe until the caller asks for it.  The reference draws its
actions from NumPy's global Mersenne twister (rollout.py:64-65); this engine draws them from the shared counter PRNG
on the device, so a rollout is a pure function of (seed, env index) -- see prng.py.

Frame dumps: PNG files (written here with zlib, no imaging library needed) or one .npy per env.  The reference encodes
mp4 through OpenCV (utility_funcs.py:28-56); `make_video` does the same when cv2 is importable and falls back to the PNG
frames when it is not -- encoding is not on the path this package accelerates.
"""
import os
import struct
import zlib

import numpy as np

from . import constants as K
from .engine import VecEngine


class Controller(object):
    """rollout.Controller (rollout.py:30-47) over a batch of envs."""

    def __init__(self, env_name="cleanup", num_envs=1, num_agents=5, seed=0, device=0, ascii_map=None):
        if env_name == "harvest":
            game = K.GAME_HARVEST
        elif env_name == "cleanup":
            game = K.GAME_CLEANUP
        else:
            raise ValueError("not a valid environment type: %r (harvest or cleanup)" % (env_name,))
        self.env_name = env_name
        # keep_beams: map_to_colors() right after a step still shows that step's beams (map_env.py:86,299-300)
        self.engine = VecEngine(game, ascii_map, num_envs=num_envs, num_agents=num_agents, seed=seed, device=device,
                                keep_beams=True)
        self.engine.reset()

    def rollout(self, horizon=50, save_path=None, envs=None):
        """`horizon` random-action steps of every env (rollout.py:49-82).

        Returns (rewards, observations, full_obs) as device tensors: rewards i32 [horizon, E, N], observations u8
        [horizon, E, N, 15, 15, 3] and full_obs u8 [horizon, count, H, W, 3] -- the frames of envs `envs` = (begin, count),
        default all.  The reference returns agent-0's rewards / observations of its one env: that is `rewards[:, 0, 0]`,
        `observations[:, 0, 0]` and `full_obs[:, 0]`.  save_path: directory that receives frameNNNNNN.png of env `begin`
        (rollout.py:74-75)."""
        import torch
        eng = self.engine
        begin, count = (0, eng.E) if envs is None else (int(envs[0]), int(envs[1]))
        dev = torch.device("cuda", eng.device)
        rewards = torch.empty((horizon, eng.E, eng.N), dtype=torch.int32, device=dev)
        n_obs = eng.E * eng.N * eng.V * eng.V * 3                               # the library wants 4-byte aligned obs blocks:
        observations = torch.empty((horizon, (n_obs + 3) & ~3), dtype=torch.uint8, device=dev)[:, :n_obs] \
            .view(horizon, eng.E, eng.N, eng.V, eng.V, 3)                       # pad the step stride when E*N is odd
        done = torch.empty((eng.E, eng.N), dtype=torch.uint8, device=dev)
        full_obs = torch.empty((horizon, count, eng.H, eng.W, 3), dtype=torch.uint8, device=dev)
        for i in range(horizon):
            eng.step_random(out=(observations[i], rewards[i], done))
            eng.render_frames(begin, count, out=full_obs[i])
        if save_path is not None:
            save_frames_png(full_obs[:, 0].cpu().numpy(), save_path)
        return rewards, observations, full_obs

    def render_rollout(self, horizon=50, path=None, fps=8, env=0):
        """rollout.py:84-114: roll out and write `<env_name>_trajectory.mp4` of env `env` under `path`."""
        if path is None:
            path = os.path.join(os.getcwd(), "videos")
        os.makedirs(path, exist_ok=True)
        _, _, full_obs = self.rollout(horizon=horizon, envs=(env, 1))
        return make_video(full_obs[:, 0].cpu().numpy(), path, video_name=self.env_name + "_trajectory", fps=fps)


def _png_bytes(rgb):
    """One uint8 [H,W,3] frame as an 8-bit truecolour PNG."""
    h, w, _ = rgb.shape
    raw = np.empty((h, 1 + 3 * w), np.uint8)
    raw[:, 0] = 0                                                             # filter type 0 on every scanline
    raw[:, 1:] = rgb.reshape(h, 3 * w)

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)) + chunk(b"IEND", b""))


def save_frames_png(frames, path, scale=1):
    """frames u8 [T,H,W,3] -> path/frame000000.png ... (the names rollout.py:75 uses); `scale` repeats every pixel
    (nearest neighbour, as the reference's resize does, utility_funcs.py:52)."""
    frames = np.asarray(frames, dtype=np.uint8)
    if frames.ndim != 4 or frames.shape[-1] != 3:
        raise ValueError("frames must be uint8 [T,H,W,3]")
    os.makedirs(path, exist_ok=True)
    names = []
    for i, f in enumerate(frames):
        if scale > 1:
            f = np.repeat(np.repeat(f, scale, axis=0), scale, axis=1)
        name = os.path.join(path, "frame" + str(i).zfill(6) + ".png")
        with open(name, "wb") as fh:
            fh.write(_png_bytes(np.ascontiguousarray(f)))
        names.append(name)
    return names


def save_frames_npy(full_obs, path, prefix="env"):
    """full_obs u8 [T,count,H,W,3] (tensor or array) -> one `<prefix>NNNNNN.npy` of shape [T,H,W,3] per env."""
    arr = full_obs.cpu().numpy() if hasattr(full_obs, "cpu") else np.asarray(full_obs)
    os.makedirs(path, exist_ok=True)
    names = []
    for e in range(arr.shape[1]):
        name = os.path.join(path, prefix + str(e).zfill(6) + ".npy")
        np.save(name, np.ascontiguousarray(arr[:, e]))
        names.append(name)
    return names


def make_video(frames, vid_path, video_name="trajectory", fps=5, resize=(640, 480)):
    """utility_funcs.make_video_from_rgb_imgs (utility_funcs.py:28-56).  Needs OpenCV, like the reference; without it
    the frames are written as PNGs under `<vid_path>/<video_name>_frames/` and that directory is returned."""
    try:
        import cv2
    except ImportError:
        return os.path.dirname(save_frames_png(frames, os.path.join(vid_path, video_name + "_frames"))[0])
    frames = np.asarray(frames, dtype=np.uint8)
    height, width = (resize[1], resize[0]) if resize is not None else frames.shape[1:3]
    out = os.path.join(vid_path, video_name + ".mp4")
    video = cv2.VideoWriter(out, cv2.VideoWriter_fourcc(*"mp4v"), float(fps), (width, height))
    for image in frames:
        if resize is not None:
            image = cv2.resize(image, resize, interpolation=cv2.INTER_NEAREST)
        video.write(image)
    video.release()
    return out
