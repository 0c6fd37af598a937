"""Frame recording for the render / record path of the reference's experiment scripts (experiments/main.py:184-186,
201-202, 255-262 use gym's `VideoRecorder(env, path=...)`, `.capture_frame()`, `.close()`).

gym and an mp4 encoder are not part of this build's environment; this recorder keeps the same three-call interface,
takes its frames from `env.render(mode='rgb_array')` (the GPU renders the RGB batch, `wurm_amd/_render.py` tiles it on
the host) and writes an animated GIF with Pillow — or, for a path ending in `.npy`, the raw (frames, H, W, 3) uint8
array.  Host side only; nothing here is on the step path.
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
        if self.path.endswith('.npy'):
            np.save(self.path, np.stack(self.frames))
        else:
            from PIL import Image
            images = [Image.fromarray(f) for f in self.frames]
            images[0].save(self.path, save_all=True, append_images=images[1:], loop=0,
                           duration=max(1, int(round(1000 / self.frames_per_sec))))
        self.frames = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
