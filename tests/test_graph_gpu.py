"""The launches are capturable in a HIP graph (torch.cuda.CUDAGraph on ROCm): the library never synchronises, allocates
or touches a stream other than the one it is given, and its one-time kernel attribute setup is done by the warm-up
call.  Small shapes (the reference's own benchmark sizes, FA2-triton.py:331-334) are launch-bound from Python; a
replayed graph removes the host cost without changing a bit of the result."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import flash_attention_impls_amd as fa  # noqa: E402


def _rand(B, H, S, D, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, H, S, D, generator=g).to(dtype).cuda() for _ in range(4)]


@pytest.mark.parametrize("causal", [False, True])
def test_forward_captured_in_a_graph_replays_bitwise(causal):
    q, k, v, _ = _rand(1, 16, 1024, 64, torch.float16, 1)            # the reference's fwd+bwd benchmark shape
    ref, ref_lse = fa.flash_attn(q, k, v, causal, return_lse=True)   # warm-up: builds nothing lazily afterwards
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fa.flash_attn(q, k, v, causal)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o, lse = fa.flash_attn(q, k, v, causal, return_lse=True)
    for _ in range(3):
        o.zero_()
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(o, ref) and torch.equal(lse, ref_lse)
    # new inputs in the captured buffers: the replay computes on them
    q2 = torch.randn_like(q)
    q.copy_(q2)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(o, fa.flash_attn(q2, k, v, causal))


def test_forward_and_backward_launches_captured_in_a_graph():
    """Forward + the three backward launches (pre-pass, dQ, dK/dV) captured on one stream through the host-side launch
    helpers the autograd Function uses (the autograd engine runs backward on a thread of its own, which is torch's
    business, not the library's: here everything is enqueued by the capturing thread)."""
    import importlib
    host = importlib.import_module("flash_attention_impls_amd.flash_attn")
    lib = fa.load_library()
    q, k, v, do = _rand(2, 4, 256, 128, torch.bfloat16, 2)
    qg, kg, vg = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o_ref = fa.flash_attn(qg, kg, vg, True)
    o_ref.backward(do)
    scale = 128 ** -0.5

    def step():
        o, lse = host._fwd_raw(lib, q, k, v, True, scale, None, True)
        return (o,) + host._bwd_raw(lib, q, k, v, o, lse, do, True, scale)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = step()
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    for got, want in zip(outs, (o_ref.detach(), qg.grad, kg.grad, vg.grad)):
        assert torch.equal(got, want)
