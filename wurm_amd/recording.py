"""Frame recording for the render / record path of the reference's experiment scripts (experiments/main.py:184-186,
201-202, 255-262 use gym's `VideoRecorder(env, path=...)`, `.capture_frame()`, `.close()`).

gym is not part of this build's environment; this recorder keeps the same three-call interface, takes its frames from
`env.render(mode='rgb_array')` (the GPU renders the RGB batch, `wurm_amd/_render.py` tiles it on the host — bit-equal
to the reference's frames, tests/test_render_golden.py) and writes, by file extension: `.mp4` through an `ffmpeg`
binary on PATH exactly as gym's recorder does (raw RGB frames piped in, libx264 / yuv420p; RuntimeError if there is no
ffmpeg, like gym's DependencyNotInstalled), `.npy` the raw (frames, H, W, 3) uint8 array, anything else an animated
GIF with Pillow.  Host side only; nothing here is on the step path.
"""
import os

import numpy as np


class VideoRecorder(object):
    def __init__(self, env, path: str, frames_per_sec: int = 12, enabled: bool = True):
        self.env = env
        self.path = path
        self.frames_per_sec = frames_per_sec
        self.enabled = enabled
        self.frames = []
        self.closed = False

    def capture_frame(self):
        """Appends the env's current frame (reference: called once per loop iteration before `model(state)`)."""
        if not self.enabled or self.closed:
            return
        frame = np.asarray(self.env.render(mode='rgb_array'))
        if frame.ndim != 3 or frame.shape[-1] != 3:
            raise RuntimeError(f'render(mode="rgb_array") returned shape {frame.shape}, expected (H, W, 3)')
        self.frames.append(frame.astype(np.uint8))

    def close(self):
        """Writes the file (nothing if no frame was captured) and releases the frames."""
        if self.closed:
            return
        self.closed = True
        if not self.enabled or not self.frames:
            self.frames = []
            return
        directory = os.path.dirname(self.path)
        if directory:
            os.makedirs(directory, exist_ok=True)
        if self.path.endswith('.mp4'):
            self._write_mp4()
        elif self.path.endswith('.npy'):
            np.save(self.path, np.stack(self.frames))
        else:
            from PIL import Image
            images = [Image.fromarray(f) for f in self.frames]
            images[0].save(self.path, save_all=True, append_images=images[1:], loop=0,
                           duration=max(1, int(round(1000 / self.frames_per_sec))))
        self.frames = []

    def _write_mp4(self):
        import shutil
        import subprocess
        ffmpeg = shutil.which('ffmpeg') or shutil.which('avconv')
        if ffmpeg is None:
            raise RuntimeError('VideoRecorder: writing .mp4 needs an ffmpeg (or avconv) binary on PATH; use a .gif or '
                               '.npy path instead')
        h, w = self.frames[0].shape[:2]
        cmd = [ffmpeg, '-nostats', '-loglevel', 'error', '-y', '-f', 'rawvideo', '-s:v', f'{w}x{h}', '-pix_fmt', 'rgb24',
               '-framerate', str(self.frames_per_sec), '-i', '-', '-vf', 'scale=trunc(iw/2)*2:trunc(ih/2)*2',
               '-vcodec', 'libx264', '-pix_fmt', 'yuv420p', '-r', str(self.frames_per_sec), self.path]
        proc = subprocess.Popen(cmd, stdin=subprocess.PIPE)
        for f in self.frames:
            proc.stdin.write(np.ascontiguousarray(f).tobytes())
        proc.stdin.close()
        if proc.wait() != 0:
            raise RuntimeError(f'VideoRecorder: {ffmpeg} failed with exit code {proc.returncode}')

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
