"""features_io.FeatureStore: the HDF5-free counterpart of the reference's features.hdf5 (one `<id>_features` record
per image) and of CocoDataLoader.get_PADDED_bboxes_batch_by_id (data/coco_dataloader.py:437-478)."""
import os

import numpy as np
import pytest
import torch

from on_device_image_captioning_amd.features_io import FeatureStore, FeatureStoreWriter


def _make(path, ids, lens, F=24):
    rng = np.random.default_rng(7)
    recs = {}
    with FeatureStoreWriter(path, F) as w:
        for i, n in zip(ids, lens):
            recs[i] = rng.standard_normal((n, F)).astype(np.float32)
            w.append(i, recs[i])
    return recs


def test_round_trip_and_reference_batch_semantics(tmp_path):
    path = str(tmp_path / "feats.odic")
    ids, lens = [391895, 7, 522418, 184613, 99], [144, 100, 144, 1, 37]
    recs = _make(path, ids, lens)
    st = FeatureStore(path)
    assert len(st) == 5 and st.img_ids == ids and st.feat_dim == 24 and 7 in st and 8 not in st
    for i in ids:
        assert torch.equal(st.get_features(i), torch.from_numpy(recs[i]))
    batch, pads = st.get_PADDED_bboxes_batch_by_id([7, 184613, 99])
    want = torch.nn.utils.rnn.pad_sequence([torch.from_numpy(recs[i]) for i in (7, 184613, 99)], batch_first=True)
    assert torch.equal(batch, want) and pads == [0, 99, 63]                  # longest in THIS batch sets the width
    batch, pads = st.get_PADDED_bboxes_batch_by_id([391895, 522418])
    assert batch.shape == (2, 144, 24) and pads == [0, 0]
    with pytest.raises(KeyError):
        st.get_PADDED_bboxes_batch_by_id([12345])


def test_writer_and_reader_reject_bad_input(tmp_path):
    path = str(tmp_path / "f.odic")
    w = FeatureStoreWriter(path, 8)
    w.append(1, np.zeros((3, 8), np.float32))
    with pytest.raises(ValueError):
        w.append(1, np.zeros((3, 8), np.float32))            # duplicate id
    with pytest.raises(ValueError):
        w.append(2, np.zeros((3, 9), np.float32))            # wrong width
    w.close()
    assert len(FeatureStore(path)) == 1
    with open(path, "r+b") as f:
        f.truncate(os.path.getsize(path) - 5)
    with pytest.raises(ValueError):
        FeatureStore(path)


@pytest.mark.gpu
def test_store_feeds_the_features_only_pipeline(tmp_path):
    """Features dumped by the HIP backbone (data_generator.py's job), read back through the pinned double-buffered
    loader and captioned by the features-only model: same captions as feeding the tensors directly."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from conftest import cached_state_dict
    from on_device_image_captioning_amd import weights as W
    from on_device_image_captioning_amd.End_ExpansionNet_v2 import make_drop_args
    from on_device_image_captioning_amd.ExpansionNet_v2 import ExpansionNet_v2
    from on_device_image_captioning_amd.pipeline import CaptionPipeline
    g, fd, dev = W.TINY, 64, "cuda:0"
    sd = cached_state_dict("TINY", "eos", end_to_end=False, img_feature_dim=fd)
    m = ExpansionNet_v2(d_model=g.d_model, N_enc=g.N_enc, N_dec=g.N_dec, ff=g.ff, num_heads=g.num_heads,
                        num_exp_enc_list=list(g.num_exp_enc_list), num_exp_dec=g.num_exp_dec,
                        output_word2idx={i: i for i in range(g.vocab_size)}, output_idx2word=list(range(g.vocab_size)),
                        max_seq_len=g.max_seq_len, drop_args=make_drop_args(), img_feature_dim=fd, rank=dev)
    m.load_state_dict(sd, strict=True)
    m.to(dev).eval()
    path = str(tmp_path / "f.odic")
    lens = [20, 17, 13, 20, 19, 8, 20, 20]
    feats = W.synth_features(8, 20, fd)
    with FeatureStoreWriter(path, fd) as w:
        for i, n in enumerate(lens):
            w.append(1000 + i, feats[i, :n])
    st = FeatureStore(path, device=dev, max_batch=4, max_tokens=20)
    pipe = CaptionPipeline(m, 4, 3, 16, 3, 2, done_poll=4, feat_len=20)
    got = []
    for lo in (0, 4):
        batch, pads = st.get_PADDED_bboxes_batch_by_id([1000 + i for i in range(lo, lo + 4)])
        assert batch.shape == (4, 20, fd) and batch.is_cuda and pads == [20 - n for n in lens[lo:lo + 4]]
        pipe.submit(batch, pads)
    while pipe.outstanding():
        got += pipe.collect()
    want = []
    for lo in (0, 4):
        x = feats[lo:lo + 4].clone()
        pads = [20 - n for n in lens[lo:lo + 4]]
        for b, n in enumerate(lens[lo:lo + 4]):
            x[b, n:] = 0.0
        toks, _ = m(enc_x=x.to(dev), enc_x_num_pads=pads, mode="beam_search", beam_size=3, how_many_outputs=1,
                    beam_max_seq_len=16, sample_or_max="max", sos_idx=3, eos_idx=2)
        want += [t[0] for t in toks]
    assert got == want
