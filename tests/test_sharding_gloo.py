"""Image-shard evaluation + the one collective of the path on CPU: two gloo ranks.

The captioner itself is replaced by a deterministic stand-in (token ids derived from the image
content) because the HIP path cannot run here; what is under test is the host logic the 8-GPU run
relies on: contiguous shards, ragged tail batches, fixed-shape all_gather, global re-ordering."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, EOS = 9, 77


def fake_caption(img_id: int):
    n = 2 + img_id % (T - 2)
    return [79] + [(img_id * 7 + j) % 1000 + 100 for j in range(n - 2)] + [EOS]


def _worker(rank, world, port, n_items, batch, out_dir):
    sys.path.insert(0, ROOT)
    from on_device_image_captioning_amd.pipeline import caption_sharded, shard_indices
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    images = torch.arange(n_items, dtype=torch.float32).view(n_items, 1, 1, 1).expand(n_items, 3, 2, 2).contiguous()

    def fetch(lo, hi):
        return images[lo:hi]

    def caption_batch(imgs):
        assert imgs.shape[0] == batch                       # ragged tails arrive padded
        toks = torch.full((batch, T), EOS, dtype=torch.int32)
        lens = torch.zeros(batch, dtype=torch.int32)
        for i in range(batch):
            c = fake_caption(int(imgs[i, 0, 0, 0]))
            toks[i, :len(c)] = torch.tensor(c, dtype=torch.int32)
            lens[i] = len(c)
        return toks, lens

    caps = caption_sharded(caption_batch, n_items, fetch, batch, T, EOS, torch.device("cpu"), rank, world)
    lo, hi, per = shard_indices(n_items, rank, world)
    torch.save({"caps": caps, "shard": (lo, hi, per)}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items,batch", [(10, 4), (7, 3), (16, 16), (3, 2)])
def test_sharded_captions_two_ranks(tmp_path, n_items, batch):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, n_items, batch, str(tmp_path)), nprocs=world, join=True)
    want = [fake_caption(i) for i in range(n_items)]
    covered = []
    for r in range(world):
        d = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert d["caps"] == want, f"rank {r}"
        lo, hi, per = d["shard"]
        covered += list(range(lo, hi))
    assert covered == list(range(n_items))               # shards are contiguous, disjoint, complete


def test_shard_indices_properties():
    sys.path.insert(0, ROOT)
    from on_device_image_captioning_amd.pipeline import shard_indices
    for n in (0, 1, 5, 16, 5000):
        for w in (1, 2, 3, 8):
            spans = [shard_indices(n, r, w) for r in range(w)]
            assert all(s[2] == spans[0][2] for s in spans)
            assert [i for lo, hi, _ in spans for i in range(lo, hi)] == list(range(n))
            assert all(hi - lo <= per for lo, hi, per in spans)


class _FakePipe:
    """Host-visible contract of CaptionPipeline that run_shard relies on (B, T, device, full / submit /
    collect_device in submission order, a bounded result ring)."""

    def __init__(self, batch, ring=3):
        self.B, self.T, self.device, self.ring, self.q = batch, T, torch.device("cpu"), ring, []
        self.max_outstanding = 0

    def full(self):
        return len(self.q) >= self.ring

    def submit(self, imgs):
        assert imgs.shape[0] == self.B and not self.full()
        self.q.append(imgs.clone())
        self.max_outstanding = max(self.max_outstanding, len(self.q))

    def collect_device(self):
        imgs = self.q.pop(0)
        toks = torch.full((self.B, T), EOS, dtype=torch.int32)
        lens = torch.zeros(self.B, dtype=torch.int32)
        for i in range(self.B):
            c = fake_caption(int(imgs[i, 0, 0, 0]))
            toks[i, :len(c)] = torch.tensor(c, dtype=torch.int32)
            lens[i] = len(c)
        return toks, lens


def _worker_run_shard(rank, world, port, n_items, batch, out_dir):
    sys.path.insert(0, ROOT)
    from on_device_image_captioning_amd.pipeline import gather_captions, run_shard, shard_indices
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi, per = shard_indices(n_items, rank, world)
    images = torch.arange(n_items, dtype=torch.float32).view(n_items, 1, 1, 1).expand(n_items, 3, 2, 2).contiguous()
    pipe = _FakePipe(batch)
    toks, lens = run_shard(pipe, hi - lo, lambda a, b: images[lo + a:lo + b], EOS)
    pt = torch.full((per, T), EOS, dtype=torch.int32)
    pl = torch.zeros(per, dtype=torch.int32)
    pt[:hi - lo], pl[:hi - lo] = toks, lens
    caps = gather_captions(pt, pl, n_items)
    torch.save({"caps": caps, "max_outstanding": pipe.max_outstanding}, os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items,batch", [(23, 4), (9, 16)])
def test_run_shard_pipelined_two_ranks(tmp_path, n_items, batch):
    """bench.py's coco5k step: the shard goes through the pipeline with several batches outstanding, ragged
    tail padded, rows land at their local index, one all_gather restores the global order."""
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_run_shard, args=(world, port, n_items, batch, str(tmp_path)), nprocs=world, join=True)
    want = [fake_caption(i) for i in range(n_items)]
    for r in range(world):
        d = torch.load(os.path.join(str(tmp_path), f"s{r}.pt"))
        assert d["caps"] == want, f"rank {r}"
        if n_items / world / batch > 3:
            assert d["max_outstanding"] == 3                # the ring was kept full


def _worker_run_steps(rank, world, port, n_steps, batch, out_dir):
    sys.path.insert(0, ROOT)
    from on_device_image_captioning_amd.pipeline import gather_captions, run_steps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # a ring of 3 distinct batches per rank; image id = 1000·rank + 10·ring slot + row
    ring = [(torch.arange(batch, dtype=torch.float32) + 1000 * rank + 10 * j).view(batch, 1, 1, 1).expand(batch, 3, 2, 2)
            .contiguous() for j in range(3)]
    pipe = _FakePipe(batch)
    toks, lens = run_steps(pipe, ring, n_steps, EOS)
    assert toks.shape == (n_steps * batch, T) and pipe.max_outstanding == min(3, n_steps) and not pipe.q
    caps = gather_captions(toks, lens, n_steps * batch * world)          # ONE collective per run
    torch.save(caps, os.path.join(out_dir, f"t{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_steps,batch", [(5, 4), (1, 16), (7, 3)])
def test_e2e16_gather_once_per_run_two_ranks(tmp_path, n_steps, batch):
    """bench.py's e2e16 step under N > 1: the token rows of all steps stay on the device and ONE all_gather at the end
    returns them rank-major, in submission order — rank r's last step is rows [(r·n + n − 1)·B, (r·n + n)·B)."""
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_run_steps, args=(world, port, n_steps, batch, str(tmp_path)), nprocs=world, join=True)
    want = [fake_caption(1000 * r + 10 * (i % 3) + b) for r in range(world) for i in range(n_steps) for b in range(batch)]
    for r in range(world):
        assert torch.load(os.path.join(str(tmp_path), f"t{r}.pt")) == want, f"rank {r}"
