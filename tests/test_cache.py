"""CPU tests of the sink + ring KV cache: bookkeeping and linearised K/V must match the
reference's SinkCacheLayer step by step (golden f8_*.npz, produced by running the reference), plus ports of the
reference's tests/test_cache.py assertions.  No GPU: the cache is host-side logic on torch tensors."""
import pytest
import torch

import golden_util as G
from sink_attention.cache import SinkAttentionCache, SinkCacheLayer


@pytest.mark.parametrize("name", G.names("f8"))
def test_cache_matches_reference_step_by_step(name):
    g = G.load(name)
    ns, W, prefill, steps, _ = (int(x) for x in g["meta"])
    k_all, v_all = g["k_all"], g["v_all"]
    state = g["state"].tolist()
    layer = SinkCacheLayer(ns, W)
    ko, vo = layer.update(k_all[:, :, :prefill], v_all[:, :, :prefill])
    assert torch.equal(ko, k_all[:, :, :prefill]) and torch.equal(vo, v_all[:, :, :prefill])   # prefill: full KV back
    assert [layer.sink_len, layer.window_len, layer.write_pos, layer.seen_tokens, ko.shape[2]] == state[0]
    lk, lv = layer.get_kv()
    assert torch.equal(lk, g["k_lin_0"]) and torch.equal(lv, g["v_lin_0"])
    twin = SinkCacheLayer(ns, W)          # the copy-free path must keep identical state
    twin.append(k_all[:, :, :prefill], v_all[:, :, :prefill])
    for i in range(steps):
        pos = prefill + i
        ko, vo = layer.update(k_all[:, :, pos:pos + 1], v_all[:, :, pos:pos + 1])
        assert [layer.sink_len, layer.window_len, layer.write_pos, layer.seen_tokens, ko.shape[2]] == state[i + 1]
        assert torch.equal(ko, g[f"k_lin_{i + 1}"]) and torch.equal(vo, g[f"v_lin_{i + 1}"])
        twin.append(k_all[:, :, pos:pos + 1], v_all[:, :, pos:pos + 1])
        assert (twin.sink_len, twin.window_len, twin.write_pos, twin.seen_tokens) == tuple(state[i + 1][:4])
        assert torch.equal(twin.window_k, layer.window_k) and torch.equal(twin.sink_v, layer.sink_v)
        # ring contents == the linearised window as a set of rows (order is irrelevant to attention)
        ring = twin.window_k[0, 0, :twin.window_len]
        lin = ko[0, 0, twin.sink_len:]
        assert sorted(map(tuple, ring.tolist())) == sorted(map(tuple, lin.tolist()))


def test_prefill_variants():                       # tests/test_cache.py:18-148
    B, H, D = 1, 4, 16
    for ns, W, N, exp in [(4, 8, 16, (4, 8, 0)), (8, 16, 4, (4, 0, 0)), (4, 8, 4, (4, 0, 0)), (4, 8, 10, (4, 6, 6)),
                          (4, 8, 20, (4, 8, 0))]:
        layer = SinkCacheLayer(ns, W)
        k, v = torch.randn(B, H, N, D), torch.randn(B, H, N, D)
        ko, vo = layer.update(k, v)
        assert ko.shape == (B, H, N, D) and torch.equal(ko, k) and torch.equal(vo, v)
        assert (layer.sink_len, layer.window_len, layer.write_pos) == exp
        if N > ns + W:     # overflow: the ring holds the newest W tokens
            assert torch.equal(layer.window_k, k[:, :, N - W:])


def test_decode_eviction_and_linearisation():      # :150-252
    B, H, D, ns, W = 1, 2, 8, 2, 4
    layer = SinkCacheLayer(ns, W)
    k = torch.arange(12, dtype=torch.float32).view(1, 1, 12, 1).expand(B, H, 12, D).contiguous()
    layer.update(k[:, :, :4], k[:, :, :4])
    for pos in range(4, 12):
        ko, _ = layer.update(k[:, :, pos:pos + 1], k[:, :, pos:pos + 1])
        ids = ko[0, 0, :, 0].tolist()
        assert ids == [0.0, 1.0] + [float(x) for x in range(max(2, pos - W + 1), pos + 1)]   # chronological
        assert layer.get_seq_length() == len(ids) and layer.get_max_cache_shape() == ns + W


def test_gqa_multi_layer_reorder_seen_tokens():    # :254-358
    cache = SinkAttentionCache(num_sink=2, window_size=4)
    B, Hkv, D = 2, 2, 8
    for layer_idx in range(3):
        k = torch.randn(B, Hkv, 6, D)
        ko, _ = cache.update(k, k, layer_idx)
        assert ko.shape == (B, Hkv, 6, D)
    assert len(cache) == 3 and cache.seen_tokens == 6 and cache.get_seq_length(0) == 6 and cache.get_max_cache_length() == 6
    k1 = torch.randn(B, Hkv, 1, D)
    ko, _ = cache.update(k1, k1, 0)
    assert ko.shape == (B, Hkv, 6, D) and cache.seen_tokens == 7
    before = cache[0].sink_k.clone()
    cache.reorder_cache(torch.tensor([1, 0]))
    assert torch.equal(cache[0].sink_k, before[[1, 0]])
    assert cache.get_seq_length(7) == 0


def test_hf_cache_isinstance():                    # :360-369
    from transformers.cache_utils import Cache
    assert isinstance(SinkAttentionCache(4, 16), Cache)
    assert SinkCacheLayer(4, 16).get_max_length() == 20


def test_decode_attention_needs_the_hip_library_not_a_fallback():
    layer = SinkCacheLayer(2, 4)
    k = torch.randn(1, 2, 5, 16)
    layer.update(k, k)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer.decode_attention(torch.randn(1, 4, 1, 16))
