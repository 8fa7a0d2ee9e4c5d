"""GPU, model level: a small random-init HuggingFace gpt-oss model run through ``patch_verl_with_sink_attention()``
against the same weights with HF's eager attention (which implements the ``sinks`` / s_aux softmax in PyTorch).
Mirrors the reference's tests/test_gpt_oss_model.py (:16-160, there on the real 20B checkpoint, which needs the
network): the kernel path must match eager and must be closer to it than attention that ignores s_aux.
No flash-attn package is needed: the patched ``_flash_attention_forward`` is the only attention that runs."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tiny_gpt_oss(dtype, seed=0):
    from transformers import GptOssConfig, GptOssForCausalLM
    torch.manual_seed(seed)
    cfg = GptOssConfig(num_hidden_layers=4, hidden_size=256, intermediate_size=256, num_attention_heads=8,
                       num_key_value_heads=2, head_dim=64, num_local_experts=2, num_experts_per_tok=2, vocab_size=512,
                       sliding_window=32, max_position_embeddings=1024,
                       layer_types=["sliding_attention", "full_attention", "sliding_attention", "full_attention"])
    cfg._attn_implementation = "eager"
    model = GptOssForCausalLM(cfg)
    with torch.no_grad():
        for layer in model.model.layers:        # make the sinks matter (they initialise near zero)
            layer.self_attn.sinks.copy_(torch.randn_like(layer.self_attn.sinks) * 2.0)
    return model.to(DEV, dtype)


def _set_impl(model, impl):
    model.config._attn_implementation_internal = impl
    for layer in model.model.layers:
        layer.self_attn.config._attn_implementation_internal = impl


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gpt_oss_logits_kernel_vs_eager(dtype):
    import sink_attention.verl_patch as vp
    from sink_attention import _native, patch_verl_with_sink_attention, unpatch_verl
    ids = torch.randint(0, 512, (2, 200), device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    with torch.no_grad():
        ref32 = _tiny_gpt_oss(torch.float32).eval()(ids).logits.float()      # fp32 eager: the yardstick
    model = _tiny_gpt_oss(dtype).eval()                                      # same seed = same weights
    with torch.no_grad():
        eager = model(ids).logits.float()
    patch_verl_with_sink_attention()
    try:
        _set_impl(model, "flash_attention_2")
        with torch.no_grad():
            out = model(ids).logits.float()
        assert ("generic" if dtype == torch.float32 else "mfma") in _native.last_path()
        # the same model with s_aux dropped at the boundary = what stock flash attention computes
        real = vp._local_s_aux
        vp._local_s_aux = lambda s_aux, H_q: None
        try:
            with torch.no_grad():
                no_aux = model(ids).logits.float()
        finally:
            vp._local_s_aux = real
    finally:
        unpatch_verl()
        _set_impl(model, "eager")
    err = (out - ref32).abs().mean().item()
    err_eager = (eager - ref32).abs().mean().item()
    err_no_aux = (no_aux - ref32).abs().mean().item()
    if dtype == torch.float32:
        assert (out - ref32).abs().max().item() < 2e-3
        assert (out.argmax(-1) == ref32.argmax(-1)).float().mean().item() > 0.999
    else:   # 16-bit: as close to the fp32 model as HF's own bf16 eager attention is
        assert err < 1.5 * err_eager + 1e-3, (err, err_eager)
    assert err * 3 < err_no_aux, (err, err_no_aux)      # handles s_aux; ignoring it is clearly worse


def test_gpt_oss_training_step_gradients_kernel_vs_eager():
    from sink_attention import patch_verl_with_sink_attention, unpatch_verl
    model = _tiny_gpt_oss(torch.float32, seed=3).train()
    ids = torch.randint(0, 512, (2, 96), device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))

    def grads():
        model.zero_grad(set_to_none=True)
        model(ids, labels=ids).loss.backward()
        layer = model.model.layers[1].self_attn
        return [p.grad.clone() for p in (layer.sinks, layer.q_proj.weight, layer.k_proj.weight, layer.v_proj.weight,
                                         model.model.layers[0].self_attn.sinks)]

    ref = grads()
    patch_verl_with_sink_attention()
    try:
        _set_impl(model, "flash_attention_2")
        got = grads()
    finally:
        unpatch_verl()
        _set_impl(model, "eager")
    for g, r in zip(got, ref):
        assert r.abs().max().item() > 0
        assert (g - r).abs().max().item() <= 2e-3 * max(1.0, r.abs().max().item()), (g - r).abs().max().item()


def _tiny_qwen2(dtype, seed=0):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    torch.manual_seed(seed)
    cfg = Qwen2Config(num_hidden_layers=2, hidden_size=256, intermediate_size=512, num_attention_heads=4,
                      num_key_value_heads=2, vocab_size=512, max_position_embeddings=1024)
    cfg._attn_implementation = "eager"
    return Qwen2ForCausalLM(cfg).to(DEV, dtype).eval()


def test_generate_with_sink_cache_end_to_end():
    """model.generate() through patch_for_generation() + SinkAttentionCache (the reference's tests/test_inference.py
    :230-292, there on a downloaded Qwen checkpoint): (1) while nothing is evicted the greedy tokens equal HF's eager
    generation; (2) with eviction every generated token equals the argmax of ONE patched prefill pass (sink + window
    mask) over the final sequence, i.e. the ring cache and the decode kernel implement the same attention pattern as
    the prefill kernel."""
    import sink_attention.generate_patch as gp
    from sink_attention import patch_for_generation, unpatch_generation
    model = _tiny_qwen2(torch.float32)
    gen = torch.Generator(device=DEV).manual_seed(5)
    ids = torch.randint(0, 512, (2, 24), device=DEV, generator=gen)
    with torch.no_grad():
        ref = model.generate(ids, max_new_tokens=12, do_sample=False)
    try:
        cache = patch_for_generation(model, num_sink=4, window_size=128)
        _set_impl(model, "flash_attention_2")
        with torch.no_grad():
            out = model.generate(ids, past_key_values=cache, max_new_tokens=12, do_sample=False)
        assert torch.equal(out, ref) and cache.seen_tokens == 24 + 11
        # eviction: 2 sinks + a window of 16 over a 36-token sequence
        cache = patch_for_generation(model, num_sink=2, window_size=16)
        with torch.no_grad():
            out2 = model.generate(ids, past_key_values=cache, max_new_tokens=12, do_sample=False)
            assert cache.get_seq_length() == 2 + 16
            logits = model(out2).logits           # one prefill pass of the patched model over the final sequence
        assert gp._GENERATION_CONFIG["window_size"] == 16
        assert torch.equal(logits[:, 23:-1].argmax(-1), out2[:, 24:])
    finally:
        unpatch_generation()
        _set_impl(model, "eager")
