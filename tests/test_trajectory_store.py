"""TrajectoryStore keeps the reference's contract that what a property returned stays valid (wurm/rl/trajectory_store.py
returns fresh torch.stack copies): a tensor read before clear() must be unchanged after the next rollout."""
import torch


def test_read_tensors_survive_clear_and_the_next_rollout():
    from wurm_amd.rl import TrajectoryStore
    store = TrajectoryStore(capacity=4)
    for t in range(5):
        store.append(reward=torch.full((3, 1), float(t)), done=torch.zeros(3, 1, dtype=torch.bool))
    kept = store.rewards            # read (handed out); dones is never read
    want = kept.clone()
    dones_buf = store._buf['done']
    store.clear()
    for t in range(5):
        store.append(reward=torch.full((3, 1), 100.0 + t), done=torch.ones(3, 1, dtype=torch.bool))
    assert torch.equal(kept, want)                       # not overwritten by the next rollout
    assert store.rewards[0, 0, 0] == 100.0
    assert store._buf['done'] is dones_buf or store._buf['done'].data_ptr() == dones_buf.data_ptr()  # unread: reused
    assert store.dones.all()
