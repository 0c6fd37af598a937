"""The record path: `wurm_amd.recording.VideoRecorder` keeps the three calls the reference's
experiments make on gym's recorder (experiments/main.py:184-186,201-202,255-262) and writes what `render('rgb_array')`
returns.  CPU: with a stand-in env; GPU: with SingleSnake / MultiSnake rendering real frames."""
import numpy as np
import pytest

from wurm_amd.recording import VideoRecorder


class _FakeEnv:
    def __init__(self):
        self.t = 0

    def render(self, mode='human'):
        assert mode == 'rgb_array'
        self.t += 1
        img = np.zeros((24, 36, 3), np.uint8)
        img[:, : self.t * 3] = (255, 40, 0)
        return img


def test_gif_and_npy(tmp_path):
    from PIL import Image
    for name in ('videos/run/0.gif', 'frames.npy'):
        rec = VideoRecorder(_FakeEnv(), path=str(tmp_path / name), frames_per_sec=10)
        for _ in range(5):
            rec.capture_frame()
        rec.close()
        rec.close()                      # idempotent
        rec.capture_frame()              # ignored after close
        if name.endswith('.gif'):
            im = Image.open(tmp_path / name)
            assert im.n_frames == 5 and im.size == (36, 24)
        else:
            arr = np.load(tmp_path / name)
            assert arr.shape == (5, 24, 36, 3) and arr.dtype == np.uint8 and arr[4, 0, 14].tolist() == [255, 40, 0]


def test_disabled_and_empty(tmp_path):
    rec = VideoRecorder(_FakeEnv(), path=str(tmp_path / 'a.gif'), enabled=False)
    rec.capture_frame()
    rec.close()
    VideoRecorder(_FakeEnv(), path=str(tmp_path / 'b.gif')).close()
    assert not (tmp_path / 'a.gif').exists() and not (tmp_path / 'b.gif').exists()


@pytest.mark.gpu
def test_records_real_envs(tmp_path):
    import torch
    from PIL import Image
    from wurm_amd.envs import MultiSnake, SingleSnake
    env = SingleSnake(num_envs=4, size=9, observation_mode='partial_2', device='cuda', seed=0)
    env.reset()
    rec = VideoRecorder(env, path=str(tmp_path / 'single.gif'))
    for t in range(6):
        rec.capture_frame()
        _, _, d, _ = env.step(torch.randint(4, (4,), device='cuda'))
        env.reset(d)
    rec.close()
    assert Image.open(tmp_path / 'single.gif').n_frames == 6
    multi = MultiSnake(2, 3, 12, device='cuda', seed=0)
    rec = VideoRecorder(multi, path=str(tmp_path / 'multi.npy'))
    rec.capture_frame()
    rec.close()
    frames = np.load(tmp_path / 'multi.npy')
    assert frames.ndim == 4 and frames.shape[0] == 1 and frames.shape[-1] == 3 and frames.max() > 0


def test_mp4_needs_ffmpeg_and_says_so(tmp_path):
    import shutil
    rec = VideoRecorder(_FakeEnv(), path=str(tmp_path / 'run.mp4'))
    rec.capture_frame()
    if shutil.which('ffmpeg') or shutil.which('avconv'):
        rec.close()
        assert (tmp_path / 'run.mp4').stat().st_size > 0
    else:
        with pytest.raises(RuntimeError, match='ffmpeg'):
            rec.close()
