"""Engine on CPU (reference ops): paged KV, chunked prefill, prefix cache, preemption,
sampling, the OpenAI server, and 2-rank tensor parallelism over gloo."""

import json
import os
import subprocess
import sys

import pytest
import torch

from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.kv_cache import BlockManager
from kvmini.engine.scheduler import SamplingParams, Scheduler, Sequence
from kvmini.engine.tokenizer import SyntheticTokenizer
from kvmini.models.llama import deinterleave_gate_up, get_config, interleave_gate_up, random_weights
from kvmini.models.reference import dense_greedy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(**kw):
    w = random_weights(get_config("tiny"), "cpu", torch.float32, seed=1)
    cfg = dict(model="tiny", device="cpu", dtype="fp32", max_num_seqs=4, max_batched_tokens=16, max_model_len=128, num_blocks=64, block_size=4)
    cfg.update(kw)
    return LLMEngine(EngineConfig(**cfg), weights=w), w


PROMPTS = [[1, 5, 9, 33, 47, 100, 21], [1] + list(range(60, 82)), [1, 5, 9, 33, 47, 100, 21, 99, 98]]


def test_padded_heads_are_the_same_model():
    """pad_head_dim: heads widened with zero rows / columns (rotate-half halves moved to the
    halves of the wider head, softmax scale and rotary width kept) give the same logits, in
    the dense oracle and through the engine's paged path."""
    from kvmini.models.llama import pad_head_dim
    from kvmini.models.reference import dense_logits

    w = random_weights(get_config("tiny"), "cpu", torch.float32, seed=3)  # head_dim 32
    wp = pad_head_dim(w, 48)
    assert wp.cfg.head_dim == 48 and wp.cfg.rope_dim == 32 and abs(wp.cfg.softmax_scale - 32 ** -0.5) < 1e-9
    assert wp.layers[0].wqkv.shape[0] == (4 + 2 * 2) * 48 and wp.layers[0].wo.shape == (128, 4 * 48)
    a, b = dense_logits(w, PROMPTS[1]), dense_logits(wp, PROMPTS[1])
    assert (a - b).abs().max().item() < 1e-4
    eng = LLMEngine(EngineConfig(model="tiny", device="cpu", dtype="fp32", max_num_seqs=4, max_batched_tokens=16, max_model_len=128, num_blocks=64, block_size=4), weights=wp)
    assert eng.generate([PROMPTS[0]], SamplingParams(max_tokens=8, ignore_eos=True))[0] == dense_greedy(w, PROMPTS[0], 8)


def test_replicated_kv_heads_are_the_same_model():
    """replicate_kv_heads: K/V heads stored twice (2 query heads per kv head become 1) give the
    same logits and the same greedy continuation through the engine."""
    from kvmini.models.llama import replicate_kv_heads
    from kvmini.models.reference import dense_logits

    w = random_weights(get_config("tiny"), "cpu", torch.float32, seed=4)  # 4 query heads, 2 kv heads
    wr = replicate_kv_heads(w, 2)
    assert wr.cfg.num_kv_heads == 4 and wr.layers[0].wqkv.shape[0] == (4 + 2 * 4) * 32
    assert (dense_logits(w, PROMPTS[1]) - dense_logits(wr, PROMPTS[1])).abs().max().item() < 1e-4
    eng = LLMEngine(EngineConfig(model="tiny", device="cpu", dtype="fp32", max_num_seqs=4, max_batched_tokens=16, max_model_len=128, num_blocks=64, block_size=4), weights=wr)
    assert eng.generate([PROMPTS[2]], SamplingParams(max_tokens=8, ignore_eos=True))[0] == dense_greedy(w, PROMPTS[2], 8)


def test_engine_matches_dense_reference_with_chunked_prefill_and_prefix_cache():
    eng, w = _engine()
    outs = eng.generate(PROMPTS, SamplingParams(max_tokens=10, ignore_eos=True))
    for p, o in zip(PROMPTS, outs):
        assert o == dense_greedy(w, p, 10)
    m = eng.metrics()
    assert m["prefix_cache_hits"] == 1 and m["kv_blocks_free"] == m["kv_blocks_total"] - 1  # block 0 is the scratch block
    assert eng.generate(PROMPTS, SamplingParams(max_tokens=10, ignore_eos=True)) == outs
    assert eng.metrics()["prefix_cache_hits"] == 4


def test_preemption_recompute_is_transparent():
    eng, w = _engine(num_blocks=14, max_num_seqs=4, enable_prefix_caching=False)
    prompts = [[1] + list(range(10 + i, 22 + i)) for i in range(4)]
    outs = eng.generate(prompts, SamplingParams(max_tokens=16, ignore_eos=True))
    assert eng.scheduler.num_preemptions > 0
    for p, o in zip(prompts, outs):
        assert o == dense_greedy(w, p, 16)


def test_eos_max_tokens_and_sampling_seed():
    eng, w = _engine()
    ref = dense_greedy(w, PROMPTS[0], 6)
    eng.scheduler.eos_token_id = ref[2]
    out = eng.generate([PROMPTS[0]], SamplingParams(max_tokens=6))[0]
    assert out == ref[:3]  # stops at (and includes) EOS
    a = eng.generate([PROMPTS[1]], SamplingParams(max_tokens=8, temperature=0.9, top_p=0.9, top_k=50, seed=7, ignore_eos=True))[0]
    b = eng.generate([PROMPTS[1]], SamplingParams(max_tokens=8, temperature=0.9, top_p=0.9, top_k=50, seed=7, ignore_eos=True))[0]
    assert a == b and len(a) == 8


def test_block_manager_refcounts_and_eviction():
    bm = BlockManager(8, 4, True)
    toks = list(range(12))
    blocks = bm.allocate(3)
    assert bm.num_free == 4 and 0 not in blocks
    assert bm.register_full_blocks(toks, blocks, 12, 0) == 3
    bm.free(blocks)
    assert bm.num_free == 7  # cached blocks stay evictable
    hit, n = bm.match_prefix(toks + [99])
    assert hit == blocks and n == 12
    hit2, n2 = bm.match_prefix(toks)  # must leave one token to compute
    assert n2 == 8
    bm.free(hit)
    bm.free(hit2)
    got = bm.allocate(7)
    assert got is not None and bm.allocate(1) is None
    assert bm.match_prefix(toks + [99]) == ([], 0)  # evicted


def test_scheduler_decode_first_and_budget():
    bm = BlockManager(64, 4, False)
    sch = Scheduler(bm, max_num_seqs=4, max_batched_tokens=8, max_model_len=64, eos_token_id=2)
    a = Sequence(1, list(range(1, 6)), SamplingParams(max_tokens=4))
    sch.add(a)
    b1 = sch.schedule()
    assert [(c.seq.req_id, c.start, c.n_tokens, c.sample) for c in b1.chunks] == [(1, 0, 5, True)]
    sch.commit(b1.chunks[0], [7])
    sch.add(Sequence(2, list(range(1, 20)), SamplingParams(max_tokens=4)))
    b2 = sch.schedule()
    assert [(c.seq.req_id, c.n_tokens, c.sample) for c in b2.chunks] == [(1, 1, True), (2, 7, False)]  # decode first, prefill chunk fills the budget


def test_tokenizer_lengths():
    tk = SyntheticTokenizer(128256, 128000, 128001)
    ids = tk.apply_chat_template([{"role": "user", "content": "Hello, world!"}])
    assert len(ids) == 14 and ids[0] == 128000 and max(ids) < 128256
    assert tk.encode("a b c") == tk.encode("a b c") and len(tk.encode("a b, c")) == 4
    assert tk.decode(tk.encode("x y")).count(" ") == 1


def test_gate_up_interleave_roundtrip():
    g, u = torch.randn(64, 8), torch.randn(64, 8)
    gu = interleave_gate_up(g, u)
    g2, u2 = deinterleave_gate_up(gu.T)
    assert torch.equal(g2.T, g) and torch.equal(u2.T, u)
    assert torch.equal(gu[:8], g[:8]) and torch.equal(gu[8:16], u[:8]) and torch.equal(gu[16:24], g[8:16])


def test_native_server_openai_and_tgi_endpoints(tmp_path):
    from kvmini.loadgen.protocols import RequestSpec
    from kvmini.loadgen.runner import LoadConfig, run_load_sync
    from kvmini.server.native import start_native_in_thread

    eng, _w = _engine(max_num_seqs=8, max_batched_tokens=32)
    _ns, url, stop = start_native_in_thread(eng, model_name="tiny")
    try:
        res = run_load_sync(LoadConfig(url=url, run_dir=str(tmp_path / "r"), requests=16, concurrency=6, spec=RequestSpec(model="tiny", max_tokens=6), quiet=True))
        assert all(r.status == 200 and r.completion_tokens == 6 and r.prompt_tokens == 14 and len(r.token_times_ms) == 6 for r in res)
        assert all(r.ttfb_ms <= r.ttft_ms for r in res)
        ns = run_load_sync(LoadConfig(url=url, requests=3, concurrency=3, spec=RequestSpec(model="tiny", max_tokens=5, stream=False), quiet=True))
        assert all(r.status == 200 and r.total_tokens == 19 for r in ns)
        tg = run_load_sync(LoadConfig(url=url, requests=3, concurrency=2, protocol="tgi", spec=RequestSpec(max_tokens=4), quiet=True))
        assert all(r.status == 200 and r.completion_tokens == 4 for r in tg)
        import urllib.request

        metrics = urllib.request.urlopen(url + "/metrics").read().decode()
        assert "vllm:generation_tokens_total" in metrics and "vllm:gpu_prefix_cache_hits_total" in metrics
        models = json.loads(urllib.request.urlopen(url + "/v1/models").read())
        assert models["data"][0]["id"] == "tiny"
        req = urllib.request.Request(url + "/v1/chat/completions", data=b'{"messages": []}', headers={"Content-Type": "application/json"})
        with pytest.raises(urllib.error.HTTPError) as ei:
            urllib.request.urlopen(req)
        assert ei.value.code == 400
    finally:
        stop()


TP_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["KV_ROOT"])
from kvmini.engine.engine import EngineConfig, LLMEngine
from kvmini.engine.scheduler import SamplingParams
from kvmini.models.llama import get_config, random_weights
from kvmini.models.reference import dense_greedy
from kvmini.parallel.telemetry_gather import TelemetryGather, table_to_power_json
dist.init_process_group("gloo")
r, ws = dist.get_rank(), dist.get_world_size()
cfg = get_config("tiny")
w = random_weights(cfg, "cpu", torch.float32, seed=5, tp_rank=r, tp_size=ws)
eng = LLMEngine(EngineConfig(model="tiny", device="cpu", dtype="fp32", max_num_seqs=4, max_batched_tokens=16, max_model_len=64, num_blocks=32, block_size=4, tp_size=ws, tp_rank=r), weights=w, tp_group=dist.group.WORLD)
prompts = [[1, 4, 8, 15, 16, 23, 42], [1, 9, 9, 9]]
if r == 0:
    # rank 0 schedules and samples; the other rank only follows the broadcast steps
    outs = eng.generate(prompts, SamplingParams(max_tokens=6, ignore_eos=True))
    eng.release_followers()
    full = random_weights(cfg, "cpu", torch.float32, seed=5)
    ok = all(o == dense_greedy(full, p, 6) for p, o in zip(prompts, outs))
else:
    served = eng.follow()
    ok = served >= 6
# lock-step telemetry: per-tick and whole-table gathers
tg = TelemetryGather()
tick = tg.gather_tick({"ts_s": 100.0 + r, "watts": 500.0 + r, "gfx_util": 50.0})
samples = [{"ts_s": 10.0 + 0.1 * i, "watts": 400.0 + 10 * r, "gfx_util": 70.0, "vram_used": 1e9} for i in range(30)]
table = tg.gather_table(samples, 10.0, 12.0, 0.5)
pj = table_to_power_json(table, 0.5)
ok = ok and tick is not None and tick.shape == (ws, 16) and list(tick[:, 0]) == list(range(ws)) and table.shape == (ws, 5, 16)
ok = ok and abs(pj["samples"][2]["watts"] - sum(400.0 + 10 * k for k in range(ws))) < 1e-9
# the native (C++) gather is all-or-nothing: a rank that cannot set it up, or whose exchange fails,
# sends EVERY rank to the torch.distributed path instead of leaving the others in a collective alone
import numpy as np
import kvmini.telemetry.native_rt as nrt
class SetupFailsOnRank1:
    def __init__(self, device_index, impl="rccl"):
        if dist.get_rank() == 1:
            raise RuntimeError("no communicator on this rank")
    def gather(self, local):
        raise AssertionError("must not be used once a rank failed to set up")
    def close(self):
        pass
class ExchangeFailsOnRank0(SetupFailsOnRank1):
    def __init__(self, device_index, impl="rccl"):
        pass
    def gather(self, local):
        if dist.get_rank() == 0:
            raise RuntimeError("exchange failed")
        return np.zeros((dist.get_world_size(),) + local.shape)  # what a lone rank would be left with
for fake in (SetupFailsOnRank1, ExchangeFailsOnRank0):
    nrt.NativeGather = fake
    tg2 = TelemetryGather(native=True)
    t2 = tg2.gather_table(samples, 10.0, 12.0, 0.5)
    ok = ok and tg2.native is None and tg2.native_error is not None and np.array_equal(np.nan_to_num(t2), np.nan_to_num(table)) and "gloo" in tg2.source
print(f"RANK{r} OK={ok}", flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


def test_tensor_parallel_and_telemetry_gather_gloo_world2(tmp_path):
    script = tmp_path / "tp.py"
    script.write_text(TP_SCRIPT)
    env = dict(os.environ, KV_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "RANK0 OK=True" in r.stdout and "RANK1 OK=True" in r.stdout


def test_speculative_decoding_matches_plain_greedy():
    """Greedy verification must reproduce the target's own greedy output exactly,
    whatever the draft proposes (random draft: ~0 accepted; identical draft: all accepted)."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights

    prompts = [[1, 5, 9, 200, 31], [1, 77], [1] + list(range(40, 75))]
    params = SamplingParams(max_tokens=12, ignore_eos=True)
    base = LLMEngine(EngineConfig(model="tiny", device="cpu", dtype="fp32", max_num_seqs=4, max_batched_tokens=32, max_model_len=128, num_blocks=64))
    want = base.generate(prompts, params)

    spec = LLMEngine(EngineConfig(model="tiny", device="cpu", dtype="fp32", max_num_seqs=4, max_batched_tokens=32, max_model_len=128, num_blocks=64, draft_model="tiny-draft", num_speculative_tokens=3))
    got = spec.generate(prompts, params)
    assert got == want
    assert spec.stats.spec_proposed > 0

    same = LLMEngine(EngineConfig(model="tiny", device="cpu", dtype="fp32", max_num_seqs=4, max_batched_tokens=32, max_model_len=128, num_blocks=64, draft_model="tiny", num_speculative_tokens=3))
    same.spec.load_draft_weights(random_weights(get_config("tiny"), same.device, torch.float32, 0))
    got2 = same.generate(prompts, params)
    assert got2 == want
    st = same.stats
    assert st.spec_accepted == st.spec_proposed > 0  # the target agrees with itself
    assert st.steps < base.stats.steps  # fewer target passes for the same tokens
    assert same.bm.num_free == same.bm.num_blocks - 1 or same.bm.num_free > 0


def test_self_speculation_early_exit_draft_is_exact_and_accepts_with_damped_tail():
    """``self:<K>`` drafts with the target's first K layers.  Output must equal plain greedy
    decoding; with the tail layers damped the draft agrees most of the time."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import LlamaConfig, random_weights

    cfg = LlamaConfig("tiny4", 512, 128, 256, 4, 4, 2, 32)
    prompts = [[1, 5, 9, 200, 31], [1, 77], [1] + list(range(40, 60))]
    params = SamplingParams(max_tokens=16, ignore_eos=True)
    common = dict(model="tiny", device="cpu", dtype="fp32", max_num_seqs=4, max_batched_tokens=32, max_model_len=128, num_blocks=64)

    def weights():
        return random_weights(cfg, "cpu", torch.float32, seed=11, tail_damping=(2, 0.02))

    base = LLMEngine(EngineConfig(**common), weights=weights())
    want = base.generate(prompts, params)
    spec = LLMEngine(EngineConfig(**common, draft_model="self:2", num_speculative_tokens=3), weights=weights())
    assert spec.spec.draft.cfg.num_layers == 2 and spec.spec.draft.w.layers[0] is spec.weights.layers[0]
    got = spec.generate(prompts, params)
    assert got == want
    st = spec.stats
    assert st.spec_proposed > 0 and st.spec_accepted / st.spec_proposed > 0.5, (st.spec_proposed, st.spec_accepted)
    # fewer target passes than tokens generated
    assert st.steps < sum(len(o) for o in got)
    # the proposals were enqueued back to back (each fed by its predecessor on the device); one read-back
    # per proposal gives the same drafts, hence the same acceptance
    slow = LLMEngine(EngineConfig(**common, draft_model="self:2", num_speculative_tokens=3), weights=weights())
    assert spec.spec.launch_ahead
    slow.spec.launch_ahead = False
    assert slow.generate(prompts, params) == want
    assert (slow.stats.spec_proposed, slow.stats.spec_accepted) == (st.spec_proposed, st.spec_accepted)


def test_batched_sampling_matches_greedy_under_top_k_1_and_respects_top_p():
    """The batched sampler (all unseeded sampled rows of a step in one call) with top_k=1 is
    greedy decoding; top_p keeps only the head of the distribution; mixed batches work."""
    from kvmini.engine.engine import EngineConfig, LLMEngine, _sample_rows
    from kvmini.engine.scheduler import SamplingParams

    common = dict(model="tiny", device="cpu", dtype="fp32", max_num_seqs=4, max_batched_tokens=32, max_model_len=128, num_blocks=64)
    prompts = [[1, 5, 9, 200, 31], [1, 77], [1, 3, 3, 7]]
    want = LLMEngine(EngineConfig(**common)).generate(prompts, SamplingParams(max_tokens=8, ignore_eos=True))
    got = LLMEngine(EngineConfig(**common)).generate(prompts, SamplingParams(max_tokens=8, ignore_eos=True, temperature=0.8, top_k=1))
    assert got == want
    torch.manual_seed(0)
    logits = torch.tensor([[5.0, 4.0, 0.0, -1.0], [0.0, 0.0, 10.0, 0.0], [1.0, 0.9, 0.5, 0.4]]).repeat(200, 1)
    params = [SamplingParams(temperature=1.0, top_p=0.6), SamplingParams(temperature=0.5), SamplingParams(temperature=1.0, top_k=2)] * 200
    toks = _sample_rows(logits, params)
    assert set(toks[0::3]) == {0}          # top_p 0.6: only the first token (p = 0.72) survives
    assert set(toks[1::3]) == {2}          # a 10-logit lead at T = 0.5 is certain
    assert set(toks[2::3]) == {0, 1}       # top_k 2 of a nearly flat row: exactly the two best ids, both drawn


def test_tensor_parallel_serving_two_ranks_gloo_matches_single_rank(tmp_path):
    """``kvmini serve --tensor-parallel-size 2`` under torchrun: rank 0 serves HTTP and
    schedules, rank 1 follows the step images from the host shared-memory channel; steps are
    launched ahead under tensor parallelism too; the greedy answers (one request, then four
    concurrent ones) equal the single-rank server's, and SIGTERM ends every rank."""
    import json
    import signal
    import time
    import urllib.request

    def ask(port):
        body = json.dumps({"model": "tiny", "messages": [{"role": "user", "content": "hello tensor parallel world"}], "max_tokens": 8}).encode()
        req = urllib.request.Request(f"http://127.0.0.1:{port}/v1/chat/completions", data=body, headers={"Content-Type": "application/json"})
        return json.loads(urllib.request.urlopen(req, timeout=30).read())["choices"][0]["message"]["content"]

    def wait_ready(port, proc):
        for _ in range(300):
            assert proc.poll() is None, "server exited early"
            try:
                urllib.request.urlopen(f"http://127.0.0.1:{port}/health", timeout=1)
                return
            except Exception:
                time.sleep(0.2)
        raise AssertionError("server not ready")

    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    common = ["--model", "tiny", "--device", "cpu", "--dtype", "fp32", "--max-model-len", "128", "--no-register"]
    tp = subprocess.Popen([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29771", "-m", "kvmini", "serve",
                           "--tensor-parallel-size", "2", "--port", "8331", *common], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    one = subprocess.Popen([sys.executable, "-m", "kvmini", "serve", "--port", "8332", *common], env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
    try:
        wait_ready(8331, tp)
        wait_ready(8332, one)
        a, b = ask(8331), ask(8332)
        assert a == b and len(a.split()) >= 4  # 8 tokens; pieces of the JSON alphabet carry no space
        # four requests in flight at once: continuous batching + launch-ahead stepping across both ranks
        from concurrent.futures import ThreadPoolExecutor

        def ask_n(args):
            port, k = args
            body = json.dumps({"model": "tiny", "messages": [{"role": "user", "content": f"request number {k} " * (k + 1)}], "max_tokens": 10 + k}).encode()
            req = urllib.request.Request(f"http://127.0.0.1:{port}/v1/chat/completions", data=body, headers={"Content-Type": "application/json"})
            return json.loads(urllib.request.urlopen(req, timeout=60).read())["choices"][0]["message"]["content"]

        with ThreadPoolExecutor(8) as ex:
            got = list(ex.map(ask_n, [(8331, k) for k in range(4)]))
            want = list(ex.map(ask_n, [(8332, k) for k in range(4)]))
        assert got == want
        info = json.loads(urllib.request.urlopen("http://127.0.0.1:8331/info", timeout=10).read())["engine"]
        assert info["pipelined_steps"] > 0, info  # steps were launched ahead under tensor parallelism
    finally:
        one.terminate()
        os.killpg(tp.pid, signal.SIGTERM)  # launcher and both ranks: nobody may be left waiting in a collective
        tp.communicate(timeout=60)
        one.wait(30)
    assert tp.returncode is not None


def test_data_parallel_router_two_cpu_replicas(tmp_path):
    """`kvmini router --replicas 2`: two engine processes behind one port; streamed and
    non-streamed requests are spread over both, metrics carry a replica label."""
    import signal
    import time
    import urllib.request

    from kvmini.loadgen.protocols import RequestSpec
    from kvmini.loadgen.runner import LoadConfig, run_load_sync

    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    p = subprocess.Popen([sys.executable, "-m", "kvmini", "router", "--replicas", "2", "--port", "8350", "--device", "cpu", "--no-register", "--", "--model", "tiny", "--dtype", "fp32", "--max-model-len", "128"],
                         env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    url = "http://127.0.0.1:8350"
    try:
        for _ in range(400):
            assert p.poll() is None, p.stdout.read()[-2000:]
            try:
                urllib.request.urlopen(url + "/health", timeout=1)
                break
            except Exception:
                time.sleep(0.2)
        res = run_load_sync(LoadConfig(url=url, requests=16, concurrency=4, spec=RequestSpec(model="tiny", max_tokens=6, stream=True), quiet=True, trace=False))
        assert all(r.status == 200 and r.completion_tokens == 6 for r in res)
        res2 = run_load_sync(LoadConfig(url=url, requests=4, concurrency=2, spec=RequestSpec(model="tiny", max_tokens=4, stream=False), quiet=True, trace=False))
        assert all(r.status == 200 and r.completion_tokens == 4 for r in res2)
        metrics = urllib.request.urlopen(url + "/metrics").read().decode()
        served = {ln.split("{")[1].split("}")[0]: float(ln.rsplit(" ", 1)[1]) for ln in metrics.splitlines() if ln.startswith("kvmini_router_requests_total{")}
        assert len(served) == 2 and all(v > 0 for v in served.values()) and sum(served.values()) == 20, served
        assert 'vllm:generation_tokens_total{replica="0"}' in metrics.replace(' ', '') or 'replica="0"' in metrics
        models = json.loads(urllib.request.urlopen(url + "/v1/models").read())
        assert models["data"][0]["id"] == "tiny"
    finally:
        os.killpg(p.pid, signal.SIGTERM)
        p.communicate(timeout=60)


def test_pipelined_stepping_equals_synchronous_stepping():
    """Launch-ahead stepping (step n+1 enqueued before step n's ids are read, token fed on the
    device) must produce the same tokens, finish reasons, prefix-cache state and block
    accounting as one-step-at-a-time, with arrivals in mid-flight, stop tokens that take effect
    one step late, an abort of a sequence whose step is in flight, and a length cap."""

    def run(async_on):
        eng, w = _engine(async_scheduling=async_on, max_num_seqs=4, max_batched_tokens=16, num_blocks=128)
        ref = dense_greedy(w, PROMPTS[0], 12)
        plan = [  # (arrival step, prompt, params)
            (0, PROMPTS[0], SamplingParams(max_tokens=12, ignore_eos=True)),
            (0, PROMPTS[1], SamplingParams(max_tokens=9, ignore_eos=True)),
            (2, PROMPTS[2], SamplingParams(max_tokens=7, ignore_eos=True, stop_token_ids=[dense_greedy(w, PROMPTS[2], 7)[3]])),
            (3, PROMPTS[0], SamplingParams(max_tokens=12, ignore_eos=True, stop_token_ids=[ref[5]])),
            (5, [1, 7, 8, 9], SamplingParams(max_tokens=30, ignore_eos=True)),  # aborted below
            (6, [1] + list(range(40, 70)), SamplingParams(max_tokens=5, ignore_eos=True)),
            (9, PROMPTS[1], SamplingParams(max_tokens=3, ignore_eos=True)),
        ]
        seqs, events, step = [], [], 0
        while step < 200:
            for at, prompt, sp in plan:
                if at == step:
                    s = eng.make_sequence(prompt, sp)
                    seqs.append(s)
                    eng.scheduler.add(s)
            if step == 9:
                eng.scheduler.abort(seqs[4].req_id)
            ev = eng.step()
            events += [(e[0], tuple(e[1]), e[2], e[3]) for e in ev]
            step += 1
            if step > 10 and not eng.scheduler.has_work() and eng._pending is None:
                break
        assert not eng.scheduler.has_work()
        m = eng.metrics()
        assert m["kv_blocks_free"] == m["kv_blocks_total"] - 1
        outs = [(s.output_ids, s.finish_reason) for i, s in enumerate(seqs) if i != 4]
        assert all(-1 not in o for o, _ in outs)
        return outs, events, eng, w

    sync_out, sync_ev, eng_s, w = run(False)
    pipe_out, pipe_ev, eng_p, _ = run(True)
    assert eng_s.pipelined_steps == 0 and eng_p.pipelined_steps > 10
    assert pipe_out == sync_out
    assert sync_out[0] == (dense_greedy(w, PROMPTS[0], 12), "length")
    assert sync_out[2][1] == "stop" and len(sync_out[2][0]) == 4
    ref = dense_greedy(w, PROMPTS[0], 12)
    assert sync_out[3][1] == "stop" and len(sync_out[3][0]) == ref.index(ref[5]) + 1
    # per request the streamed events carry the same tokens in the same order
    for rid in {e[0] for e in sync_ev}:
        if rid == 5:
            continue
        assert [e[1:] for e in pipe_ev if e[0] == rid] == [e[1:] for e in sync_ev if e[0] == rid]


def test_seeded_requests_and_requests_with_logprobs_stay_launched_ahead():
    """A seeded request's variates are a function of (seed, token index), so its rows are sampled
    by the device ops of a step that was launched ahead: same tokens as the synchronous path, step
    by step, whatever else shares the batch.  Logprobs are extracted by device ops behind the pick
    and reach the host with the step's ids: the same values as the synchronous path reads off the
    logits, without leaving the launch-ahead."""
    from kvmini.engine.engine import seeded_uniform

    us = [seeded_uniform(7, k) for k in range(2000)]
    assert all(0.0 <= u < 1.0 for u in us) and abs(sum(us) / len(us) - 0.5) < 0.03 and len(set(us)) > 1990
    assert seeded_uniform(7, 3) == us[3] and seeded_uniform(8, 3) != us[3]
    outs = {}
    for mode in (True, False):
        eng, w = _engine(max_num_seqs=4, async_scheduling=mode)
        a = eng.make_sequence(PROMPTS[0], SamplingParams(max_tokens=10, ignore_eos=True))
        eng.scheduler.add(a)
        for _ in range(4):
            eng.step()
        if mode:
            assert eng._pending is not None and eng.pipelined_steps == 4
        b = eng.make_sequence(PROMPTS[1], SamplingParams(max_tokens=9, temperature=0.8, top_p=0.95, seed=3, ignore_eos=True))
        c = eng.make_sequence(PROMPTS[1], SamplingParams(max_tokens=9, temperature=0.8, top_p=0.95, seed=4, ignore_eos=True))
        eng.scheduler.add(b)
        eng.scheduler.add(c)
        before = eng.pipelined_steps
        while eng.scheduler.has_work():
            eng.step()
        if mode:
            assert eng.pipelined_steps - before >= 9  # the seeded rows did not stop the launch-ahead
        assert eng._pending is None and a.output_ids == dense_greedy(w, PROMPTS[0], 10) and -1 not in b.output_ids
        outs[mode] = (b.output_ids, c.output_ids)
    assert outs[True] == outs[False] and outs[True][0] != outs[True][1] and len(outs[True][0]) == 9
    # alone in the batch: the same tokens again
    eng, _w = _engine(max_num_seqs=4)
    assert eng.generate([PROMPTS[1]], SamplingParams(max_tokens=9, temperature=0.8, top_p=0.95, seed=3, ignore_eos=True))[0] == outs[True][0]
    # logprobs: greedy and sampled (seeded) requests, launched ahead and synchronous
    logs = {}
    for mode in (True, False):
        eng, _w = _engine(max_num_seqs=4, async_scheduling=mode)
        a = eng.make_sequence(PROMPTS[0], SamplingParams(max_tokens=10, ignore_eos=True))
        eng.scheduler.add(a)
        for _ in range(4):
            eng.step()
        if mode:
            assert eng._pending is not None
        d = eng.make_sequence(PROMPTS[1], SamplingParams(max_tokens=6, logprobs=3, ignore_eos=True))
        e = eng.make_sequence(PROMPTS[2], SamplingParams(max_tokens=6, logprobs=2, temperature=0.7, seed=5, ignore_eos=True))
        eng.scheduler.add(d)
        eng.scheduler.add(e)
        before = eng.pipelined_steps
        while eng.scheduler.has_work():
            eng.step()
        if mode:
            assert eng.pipelined_steps - before >= 6
        assert eng._pending is None and len(d.logprob_log) == 6 and len(e.logprob_log) == 6
        logs[mode] = (d.output_ids, e.output_ids, d.logprob_log, e.logprob_log)
    assert logs[True][0] == logs[False][0] and logs[True][1] == logs[False][1]
    for la, lb in zip(logs[True][2] + logs[True][3], logs[False][2] + logs[False][3]):
        assert abs(la[0] - lb[0]) < 1e-4 and [i for i, _ in la[1]] == [i for i, _ in lb[1]]
        assert all(abs(x - y) < 1e-4 for (_, x), (_, y) in zip(la[1], lb[1]))
    assert all(len(top) == 3 for _lp, top in logs[True][2]) and all(len(top) == 2 for _lp, top in logs[True][3])
    assert all(lp == top[0][1] for lp, top in logs[True][2])  # greedy: the picked token is the most likely one


def test_gemm_workgroup_planner_fills_the_cus_evenly():
    """Pure-Python part of the streaming-GEMM dispatch: waves per workgroup and K splits are
    chosen so that the grid is a whole number of rounds over the 256 CUs where the shape allows."""
    from kvmini.ops.hip_ops import MAX_SPLITS, _plan_slabs, _waves_per_workgroup

    assert _waves_per_workgroup(28672) == 7  # 1792 tiles = 7 x 256
    assert _waves_per_workgroup(57344) == 7
    assert _waves_per_workgroup(128256) == 8  # 4 and 8 waves fill alike: the larger workgroup wins
    assert _waves_per_workgroup(16384) == 4  # 256 workgroups of 4 waves beat 128 of 8
    for n, k in ((6144, 4096), (4096, 4096), (4096, 14336), (8192, 8192), (8192, 28672), (10240, 8192), (3072, 2048), (2048, 8192), (5120, 13824), (256, 256), (4096, 11008)):
        nw, splits, gps = _plan_slabs(n, k)
        assert nw in (4, 6, 7, 8) and n % (16 * nw) == 0
        assert 1 <= splits <= MAX_SPLITS and splits * gps >= k // 256 > (splits - 1) * gps  # the splits cover K, the last one may be shorter
    for n, k in ((6144, 4096), (4096, 4096), (4096, 14336)):  # the Llama-3-8B projections: one workgroup per CU
        nw, splits, _ = _plan_slabs(n, k)
        assert n // (16 * nw) * splits == 256
    # decode attention: 8 waves per workgroup only while (tokens x kv heads) leaves CUs idle
    from kvmini.ops.hip_ops import _decode_attention_waves

    assert _decode_attention_waves(20, 8) == 8 and _decode_attention_waves(32, 8) == 8
    assert _decode_attention_waves(10, 32) == 4 and _decode_attention_waves(64, 8) == 4


def test_unseeded_sampling_rides_the_pipeline():
    """temperature > 0 without a seed is sampled by device ops behind the forward pass, so the
    step after it can still be launched ahead.  top_k = 1 makes sampling deterministic: the
    tokens must equal greedy decoding; a free sample must stay in the vocabulary."""
    eng, w = _engine(max_num_seqs=4)
    a = eng.make_sequence(PROMPTS[0], SamplingParams(max_tokens=10, temperature=0.7, top_k=1, ignore_eos=True))
    b = eng.make_sequence(PROMPTS[1], SamplingParams(max_tokens=8, ignore_eos=True))
    c = eng.make_sequence(PROMPTS[2], SamplingParams(max_tokens=9, temperature=1.3, top_p=0.8, ignore_eos=True))
    for s in (a, b, c):
        eng.scheduler.add(s)
    steps = 0
    while eng.scheduler.has_work():
        eng.step()
        steps += 1
    assert eng.pipelined_steps == steps - 1  # every step but the final read-back launched one
    assert a.output_ids == dense_greedy(w, PROMPTS[0], 10)
    assert b.output_ids == dense_greedy(w, PROMPTS[1], 8)
    assert len(c.output_ids) == 9 and all(0 <= t < w.cfg.vocab_size for t in c.output_ids)


def test_native_server_n_choices_stop_strings_and_logprobs():
    """OpenAI request fields beyond the basics: ``n`` choices (n sequences sharing the prompt),
    ``stop`` strings (text-level: the engine only knows stop tokens), ``logprobs`` /
    ``top_logprobs`` in both dialects, streamed and not."""
    import math
    import urllib.error
    import urllib.request

    from kvmini.server.native import start_native_in_thread

    eng, _w = _engine(max_num_seqs=8, max_batched_tokens=32)
    ns, url, stop = start_native_in_thread(eng, model_name="tiny")
    eng_text = lambda ids: ns.tok.decode(ids)  # noqa: E731

    def post(path, body):
        req = urllib.request.Request(url + path, data=json.dumps(body).encode(), headers={"Content-Type": "application/json"})
        return urllib.request.urlopen(req, timeout=60).read().decode()

    try:
        msgs = [{"role": "user", "content": "Hello, world!"}]
        base = json.loads(post("/v1/chat/completions", {"model": "tiny", "messages": msgs, "max_tokens": 8}))
        text = base["choices"][0]["message"]["content"]
        assert base["usage"]["completion_tokens"] == 8 and base["choices"][0]["finish_reason"] == "length"
        # n = 3 greedy choices: three identical answers, usage counts all of them
        r = json.loads(post("/v1/chat/completions", {"model": "tiny", "messages": msgs, "max_tokens": 8, "n": 3}))
        assert [c["index"] for c in r["choices"]] == [0, 1, 2] and all(c["message"]["content"] == text for c in r["choices"])
        assert r["usage"]["completion_tokens"] == 24
        # a stop string cuts the text in front of the match
        words = text.split(" ")
        cut = next(w for w in words[2:] if w)
        r = json.loads(post("/v1/chat/completions", {"model": "tiny", "messages": msgs, "max_tokens": 8, "stop": [cut]}))
        assert r["choices"][0]["finish_reason"] == "stop" and r["choices"][0]["message"]["content"] == text[: text.find(cut)]
        # logprobs: one entry per token, the greedy token is the top one, probabilities are sane
        r = json.loads(post("/v1/chat/completions", {"model": "tiny", "messages": msgs, "max_tokens": 5, "logprobs": True, "top_logprobs": 3}))
        content = r["choices"][0]["logprobs"]["content"]
        assert len(content) == 5 and all(len(e["top_logprobs"]) == 3 for e in content)
        assert all(e["logprob"] <= 0 and abs(e["logprob"] - e["top_logprobs"][0]["logprob"]) < 1e-5 for e in content)
        assert all(sum(math.exp(t["logprob"]) for t in e["top_logprobs"]) <= 1.0 + 1e-4 for e in content)
        r = json.loads(post("/v1/completions", {"model": "tiny", "prompt": "abc", "max_tokens": 4, "logprobs": 2}))
        lp = r["choices"][0]["logprobs"]
        assert len(lp["tokens"]) == 4 and len(lp["token_logprobs"]) == 4 and all(len(d) <= 2 for d in lp["top_logprobs"])
        # streamed: n = 2 with logprobs, chunks carry the choice index
        raw = post("/v1/chat/completions", {"model": "tiny", "messages": msgs, "max_tokens": 4, "n": 2, "stream": True, "logprobs": True, "stream_options": {"include_usage": True}})
        events = [json.loads(ln[6:]) for ln in raw.splitlines() if ln.startswith("data: ") and ln != "data: [DONE]"]
        per_choice = {0: "", 1: ""}
        for ev in events:
            for c in ev["choices"]:
                per_choice[c["index"]] += (c.get("delta") or {}).get("content") or ""
        assert per_choice[0] == per_choice[1] != "" and events[-1]["usage"]["completion_tokens"] == 8
        assert sum(1 for ev in events for c in ev["choices"] if c.get("logprobs")) == 8
        # logit_bias (string keys, as JSON has them): +100 on one token makes it every pick; a bad id is the client's error
        r = json.loads(post("/v1/completions", {"model": "tiny", "prompt": "abc", "max_tokens": 4, "logit_bias": {"65": 100}, "logprobs": 1}))
        assert r["choices"][0]["logprobs"]["tokens"] == [r["choices"][0]["logprobs"]["tokens"][0]] * 4
        assert r["choices"][0]["text"] == eng_text([65] * 4)
        for bad in ({"logit_bias": {"x": 1}}, {"logit_bias": {"99999999": 1}}, {"min_tokens": 50}):
            try:
                post("/v1/completions", {"model": "tiny", "prompt": "abc", "max_tokens": 4, **bad})
                raise AssertionError(f"accepted {bad}")
            except urllib.error.HTTPError as e:
                assert e.code == 400, (bad, e.code)
        r = json.loads(post("/v1/completions", {"model": "tiny", "prompt": "abc", "max_tokens": 6, "min_tokens": 6, "ignore_eos": False}))
        assert r["usage"]["completion_tokens"] == 6
        # /tokenize and /detokenize: the ids a completion request would be fed, and back
        tk = json.loads(post("/tokenize", {"prompt": "Hello, world!"}))
        r = json.loads(post("/v1/completions", {"model": "tiny", "prompt": "Hello, world!", "max_tokens": 1}))
        assert tk["count"] == len(tk["tokens"]) == r["usage"]["prompt_tokens"] and tk["max_model_len"] == eng.ecfg.max_model_len
        assert json.loads(post("/tokenize", {"prompt": "Hello, world!", "add_special_tokens": False}))["tokens"] == tk["tokens"][1:]
        assert json.loads(post("/tokenize", {"messages": msgs}))["count"] == base["usage"]["prompt_tokens"]
        assert json.loads(post("/detokenize", {"tokens": tk["tokens"][1:]}))["prompt"] == eng_text(tk["tokens"][1:])
        for path, bad in (("/tokenize", {}), ("/detokenize", {"tokens": [10 ** 9]}), ("/detokenize", {"tokens": "x"})):
            try:
                post(path, bad)
                raise AssertionError(f"accepted {bad}")
            except urllib.error.HTTPError as e:
                assert e.code == 400
        assert eng.metrics()["running"] == 0
    finally:
        stop()


def test_ride_along_budget_keeps_bursts_in_streaming_sized_steps():
    """With ``ride_along_limit`` a small prompt backlog is fed to the running decodes in steps of
    at most 64 tokens; a long prompt still gets the full budget; outputs are unchanged."""
    eng, w = _engine(max_num_seqs=16, max_batched_tokens=256, max_model_len=512, num_blocks=400, ride_along_limit=128)
    sch = eng.scheduler
    first = [eng.make_sequence([1, 2 + i, 3, 4], SamplingParams(max_tokens=30, ignore_eos=True)) for i in range(3)]
    for s in first:
        sch.add(s)
    eng.step(); eng.step(); eng.step()
    burst = [eng.make_sequence([1] + list(range(10 + i, 40 + i)), SamplingParams(max_tokens=4, ignore_eos=True)) for i in range(4)]  # 4 x 31 = 124 tokens
    for s in burst:
        sch.add(s)
    assert sch._step_budget() == 64
    sizes = []
    while any(s.is_prefilling for s in burst):
        b = sch._step_budget()
        eng.step()
        sizes.append(b)
    assert sizes and max(sizes) == 64 and len(sizes) >= 2
    long = eng.make_sequence([1] + [(7 * i) % 200 + 5 for i in range(300)], SamplingParams(max_tokens=2, ignore_eos=True))
    sch.add(long)
    assert sch._step_budget() == 256  # backlog above the limit: full budget
    while sch.has_work():
        eng.step()
    for s in burst:
        assert s.output_ids == dense_greedy(w, s.prompt_ids, 4)
    assert long.output_ids == dense_greedy(w, long.prompt_ids, 2)


def test_malformed_requests_fail_alone_and_early():
    """A request the engine can never serve is refused at intake (HTTP 400 / 422 / 401) and the
    requests running next to it finish normally: bad seed, non-finite temperature, a prompt at
    the model length, a request larger than the whole KV cache — on the OpenAI and TGI routes,
    which also both check the API key."""
    import threading
    import urllib.error
    import urllib.request

    from kvmini.server.native import start_native_in_thread

    eng, _w = _engine(max_num_seqs=4, max_batched_tokens=32, max_model_len=64, num_blocks=12, block_size=4)
    _ns, url, stop = start_native_in_thread(eng, model_name="tiny", api_key="k")

    def post(path, body, key="k"):
        req = urllib.request.Request(url + path, data=json.dumps(body).encode(), headers={"Content-Type": "application/json", "Authorization": f"Bearer {key}"})
        try:
            with urllib.request.urlopen(req, timeout=60) as r:
                return r.status, json.loads(r.read().decode())
        except urllib.error.HTTPError as e:
            return e.code, json.loads(e.read().decode() or "{}")

    try:
        msgs = [{"role": "user", "content": "hi"}]
        good: dict = {}
        th = threading.Thread(target=lambda: good.update(r=post("/v1/chat/completions", {"model": "tiny", "messages": msgs, "max_tokens": 24})))
        th.start()
        assert post("/v1/chat/completions", {"model": "tiny", "messages": msgs, "max_tokens": 4, "temperature": 0.7, "seed": "abc"})[0] == 400
        assert post("/v1/chat/completions", {"model": "tiny", "messages": msgs, "max_tokens": 4, "temperature": 0.7, "seed": 1.5})[0] == 400
        assert post("/v1/chat/completions", {"model": "tiny", "messages": msgs, "max_tokens": 4, "temperature": float("inf")})[0] == 400
        long_text = "word " * 200
        st, body = post("/v1/chat/completions", {"model": "tiny", "messages": [{"role": "user", "content": long_text}], "max_tokens": 4})
        assert st == 400 and "max_model_len" in body["error"]["message"]
        # fits the model length but not the 11 usable KV blocks of 4 tokens
        st, body = post("/v1/completions", {"model": "tiny", "prompt": "a b c", "max_tokens": 60})
        assert st == 400 and "KV blocks" in body["error"]["message"]
        # TGI routes: key checked, same limits
        assert post("/generate", {"inputs": "hello", "parameters": {"max_new_tokens": 4}}, key="wrong")[0] == 401
        assert post("/generate", {"inputs": long_text, "parameters": {"max_new_tokens": 4}})[0] == 400
        assert post("/generate", {"inputs": "hello", "parameters": {"max_new_tokens": 4, "do_sample": True, "temperature": 0.5, "seed": "x"}})[0] == 422
        st, body = post("/generate", {"inputs": "hello", "parameters": {"max_new_tokens": 4}})
        assert st == 200 and body["details"]["generated_tokens"] == 4
        th.join(60)
        assert good["r"][0] == 200 and good["r"][1]["usage"]["completion_tokens"] == 24
        assert eng.metrics()["running"] == 0 and eng.metrics()["waiting"] == 0
    finally:
        stop()
    # the engine's own intake refuses too (a front-end that skipped the check)
    with pytest.raises(ValueError):
        eng.generate([[1] * 64], SamplingParams(max_tokens=4))


def test_llama3_rope_scaling_tables_and_checkpoint_config_check(tmp_path):
    """The 3.1 / 3.2 presets carry the 'llama3' rotary scaling: high-frequency pairs are those of
    the plain table, low-frequency pairs turn `factor` times slower; a checkpoint whose
    config.json disagrees with the preset is refused."""
    from kvmini.models.llama import check_hf_config, rope_tables

    c31, c3 = get_config("llama-3.1-8b"), get_config("llama-3-8b")
    cos31, sin31 = rope_tables(c31, 64, "cpu")
    cos3, sin3 = rope_tables(c3, 64, "cpu")
    assert torch.equal(cos31[:, :16], cos3[:, :16])  # wavelengths far below the original context: untouched
    ang3 = torch.atan2(sin3[1, -1].double(), cos3[1, -1].double())
    ang31 = torch.atan2(sin31[1, -1].double(), cos31[1, -1].double())
    assert abs(float(ang3 / ang31) - 8.0) < 1e-3  # the slowest pair: 8x slower
    assert get_config("meta-llama/Llama-3.1-8B-Instruct").rope_scaling == (8.0, 1.0, 4.0, 8192)
    good = {"hidden_size": 4096, "intermediate_size": 14336, "num_hidden_layers": 32, "num_attention_heads": 32, "num_key_value_heads": 8, "vocab_size": 128256,
            "rms_norm_eps": 1e-5, "rope_theta": 500000.0, "tie_word_embeddings": False,
            "rope_scaling": {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}}
    p = tmp_path / "config.json"
    p.write_text(json.dumps(good))
    check_hf_config(str(p), c31)
    with pytest.raises(ValueError, match="rope_scaling"):
        check_hf_config(str(p), c3)
    p.write_text(json.dumps({**good, "num_key_value_heads": 4}))
    with pytest.raises(ValueError, match="num_key_value_heads"):
        check_hf_config(str(p), c31)


def test_rejection_sampling_reproduces_the_target_distribution():
    """Monte Carlo over the verifier alone: whatever the draft distribution is, the first emitted
    token is distributed as the target's p_0 and, given that the first draft was kept, the second
    as p_1 (Leviathan et al. 2023, theorem 1)."""
    from kvmini.engine.speculative import rejection_sample

    g = torch.Generator().manual_seed(5)
    V, k, N = 6, 2, 30000
    p = torch.softmax(torch.randn(k + 1, V, generator=g) * 1.5, -1)
    q = torch.softmax(torch.randn(k, V, generator=g) * 1.5, -1)
    q[0, 2] = 0.0  # a token the draft never proposes
    q[0] /= q[0].sum()
    first = torch.zeros(V)
    second = torch.zeros(V)
    n_second = 0
    accepted = 0
    for _ in range(N):
        drafts = torch.stack([torch.multinomial(q[i], 1, generator=g).view(()) for i in range(k)])
        a, tok = rejection_sample(p, q, drafts, g)
        a, tok = int(a), int(tok)
        out = drafts[:a].tolist() + [tok]
        first[out[0]] += 1
        accepted += a
        if a >= 1:
            second[out[1]] += 1
            n_second += 1
    assert (first / N - p[0]).abs().sum().item() / 2 < 0.015  # total variation
    assert (second / n_second - p[1]).abs().sum().item() / 2 < 0.02
    # expected acceptance of the first draft = sum_x min(p_0, q_0)
    assert abs(n_second / N - torch.minimum(p[0], q[0]).sum().item()) < 0.015
    # identical distributions: everything is accepted and the bonus token comes from p_k
    a, tok = rejection_sample(p, p[:k].clone(), torch.tensor([1, 3]), g)
    assert int(a) == k and 0 <= int(tok) < V


def test_speculative_decoding_with_temperature_uses_rejection_sampling():
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights

    common = dict(model="tiny", device="cpu", dtype="fp32", max_num_seqs=4, max_batched_tokens=64, max_model_len=128, num_blocks=64)
    prompts = [[1, 5, 9, 200, 31], [1, 77, 4]]
    eng = LLMEngine(EngineConfig(**common, draft_model="tiny-draft", num_speculative_tokens=3))
    out = eng.generate(prompts, SamplingParams(max_tokens=20, ignore_eos=True, temperature=0.9, top_p=0.95, seed=7))
    assert [len(o) for o in out] == [20, 20] and eng.stats.spec_proposed > 0
    again = LLMEngine(EngineConfig(**common, draft_model="tiny-draft", num_speculative_tokens=3)).generate(prompts, SamplingParams(max_tokens=20, ignore_eos=True, temperature=0.9, top_p=0.95, seed=7))
    assert again == out  # seeded requests are reproducible through the speculative path too
    # a draft that IS the target: p == q at every position, so every draft token is kept
    same = LLMEngine(EngineConfig(**common, draft_model="tiny", num_speculative_tokens=3))
    same.spec.load_draft_weights(random_weights(get_config("tiny"), "cpu", torch.float32, same.ecfg.seed))
    out = same.generate(prompts, SamplingParams(max_tokens=21, ignore_eos=True, temperature=1.0))
    assert [len(o) for o in out] == [21, 21]
    assert same.stats.spec_accepted == same.stats.spec_proposed > 0
    assert same.stats.steps < 21  # far fewer target passes than tokens
    # greedy and sampling requests side by side in one engine
    mixed = LLMEngine(EngineConfig(**common, draft_model="tiny-draft", num_speculative_tokens=2))
    seqs = [mixed.make_sequence(prompts[0], SamplingParams(max_tokens=10, ignore_eos=True)), mixed.make_sequence(prompts[1], SamplingParams(max_tokens=10, ignore_eos=True, temperature=0.7))]
    for s_ in seqs:
        assert mixed.scheduler.add(s_) is None
    while mixed.scheduler.has_work():
        mixed.step()
    plain = LLMEngine(EngineConfig(**common)).generate([prompts[0]], SamplingParams(max_tokens=10, ignore_eos=True))[0]
    assert seqs[0].output_ids == plain and len(seqs[1].output_ids) == 10


def test_prefix_cache_serves_a_block_only_for_equal_tokens_behind_an_equal_chain():
    """The hash key is a lookup, not a proof: a block is re-used only when its stored tokens and parent chain
    equal the request's (a forged or accidental key collision is a miss), chains registered in pieces continue,
    and an evicted block's content record goes with it."""
    from kvmini.engine.kv_cache import BlockManager

    bm = BlockManager(64, 16)
    a = list(range(100, 132))
    blk = bm.allocate(2)
    assert bm.register_full_blocks(a, blk, 32, 0) == 2
    other = list(range(500, 532))
    bm._cached[bm.chain_hash(0, other[:16])] = blk[0]  # as if the keys of two different blocks collided
    got, n = bm.match_prefix(other + [1])
    assert (got, n) == ([], 0)
    # same second block behind a different first block: the parent chain differs, no hit beyond what is equal
    mixed = other[:16] + a[16:]
    assert bm.match_prefix(mixed + [1]) == ([], 0)
    got, n = bm.match_prefix(a + [1])
    assert n == 32 and got == blk
    bm.free(got)
    # registered in two pieces, found as one chain
    bm2 = BlockManager(64, 16)
    bl, t4 = bm2.allocate(4), list(range(64))
    n1 = bm2.register_full_blocks(t4, bl, 32, 0)
    assert bm2.register_full_blocks(t4, bl, 64, n1) == 4
    got, n = bm2.match_prefix(t4 + [9])
    assert n == 64 and got == bl
    bm2.free(got)
    bm2.free(bl)
    assert len(bm2._content) == 4
    taken = bm2.allocate(63)  # evicts every cached block
    assert taken is not None and not bm2._content and not bm2._cached


def test_qkv_bias_models_match_the_dense_oracle():
    """Qwen2 form (a bias on the fused Q / K / V projection): the engine's tokens equal the dense fp32 forward pass,
    and the bias survives head padding and K/V head replication."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, pad_head_dim, random_weights, replicate_kv_heads
    from kvmini.models.reference import dense_greedy

    w = random_weights(get_config("tiny-qkv-bias"), "cpu", torch.float32, seed=4)
    assert w.layers[0].bqkv is not None and w.layers[0].bqkv.numel() == w.layers[0].wqkv.shape[0]
    for lw in w.layers:
        lw.bqkv.mul_(20.0)  # (N(0, 0.02) draws: make the bias matter)
    prompts = [[1, 7, 9, 30, 2, 44], [1, 100, 101]]
    want = [dense_greedy(w, p, 8) for p in prompts]
    eng = LLMEngine(EngineConfig(model="tiny-qkv-bias", device="cpu", dtype="fp32", kernels="torch", max_model_len=64, max_num_seqs=4, max_batched_tokens=32, use_graphs=False), weights=w)
    assert eng.generate(prompts, SamplingParams(max_tokens=8, ignore_eos=True)) == want
    zero = random_weights(get_config("tiny-qkv-bias"), "cpu", torch.float32, seed=4)
    for lw in zero.layers:
        lw.bqkv.zero_()
    assert [dense_greedy(zero, p, 8) for p in prompts] != want
    wide = pad_head_dim(w, 64)
    assert wide.layers[0].bqkv.numel() == wide.layers[0].wqkv.shape[0] and [dense_greedy(wide, p, 8) for p in prompts] == want
    rep = replicate_kv_heads(w, 2)
    assert rep.layers[0].bqkv.numel() == rep.layers[0].wqkv.shape[0] and [dense_greedy(rep, p, 8) for p in prompts] == want


def _penalised_greedy(w, prompt, n, presence=0.0, frequency=0.0, repetition=1.0):
    """Reference: greedy decoding with OpenAI's presence / frequency penalties over the generated tokens and the
    multiplicative repetition penalty over prompt + generated tokens, on the dense fp32 forward pass."""
    from kvmini.models.reference import dense_logits

    ids, out = list(prompt), []
    for _ in range(n):
        lg = dense_logits(w, ids)[-1].clone()
        cnt = torch.zeros_like(lg)
        for t in out:
            cnt[t] += 1
        if repetition != 1.0:
            seen = cnt > 0
            seen[torch.tensor(prompt)] = True
            lg = torch.where(seen, torch.where(lg > 0, lg / repetition, lg * repetition), lg)
        lg = lg - frequency * cnt - presence * (cnt > 0).float()
        t = int(lg.argmax())
        out.append(t)
        ids.append(t)
    return out


@pytest.mark.parametrize("async_scheduling", [True, False])
def test_sampling_penalties_match_the_reference(async_scheduling):
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights
    from kvmini.models.reference import dense_greedy

    w = random_weights(get_config("tiny"), "cpu", torch.float32, seed=2)
    eng = LLMEngine(EngineConfig(model="tiny", device="cpu", dtype="fp32", kernels="torch", max_model_len=128, max_num_seqs=4, max_batched_tokens=32, use_graphs=False,
                                 async_scheduling=async_scheduling), weights=w)
    prompts = [[1, 7, 9, 30, 2, 44], [1, 100, 101, 100, 101, 100]]
    plain = [dense_greedy(w, p, 24) for p in prompts]
    cases = [dict(frequency_penalty=1.5), dict(presence_penalty=2.0), dict(repetition_penalty=1.8), dict(presence_penalty=0.5, frequency_penalty=0.7, repetition_penalty=1.2)]
    for kw in cases:
        ref_kw = dict(presence=kw.get("presence_penalty", 0.0), frequency=kw.get("frequency_penalty", 0.0), repetition=kw.get("repetition_penalty", 1.0))
        want = [_penalised_greedy(w, p, 24, **ref_kw) for p in prompts]
        # one penalised and one plain request in the same steps: the plain one is untouched
        outs = eng.generate(prompts + [prompts[0]], [SamplingParams(max_tokens=24, ignore_eos=True, **kw)] * 2 + [SamplingParams(max_tokens=24, ignore_eos=True)])
        assert outs[:2] == want, (kw, outs[:2], want)
        assert outs[2] == plain[0]
    assert any(_penalised_greedy(w, prompts[0], 24, frequency=1.5) != plain[0] for _ in range(1))  # (the penalty changes this model's output at all)
    # sampled, seeded: the same tokens twice, and different from the unpenalised draw
    sp = SamplingParams(max_tokens=16, ignore_eos=True, temperature=0.9, seed=3, frequency_penalty=1.0, presence_penalty=1.0)
    a, b = eng.generate([prompts[0]], sp)[0], eng.generate([prompts[0]], sp)[0]
    assert a == b and len(a) == 16
    # out-of-range values are refused at intake
    bad = eng.make_sequence(prompts[0], SamplingParams(max_tokens=4, presence_penalty=9.0))
    assert "presence_penalty" in (eng.scheduler.validate(bad) or "")


@pytest.mark.parametrize("async_scheduling", [True, False])
def test_logit_bias_and_min_tokens(async_scheduling):
    """``logit_bias`` adds to the named logits before the pick; ``min_tokens`` keeps EOS and the stop tokens from being
    picked before that many tokens exist — both against the dense fp32 forward pass with the same rule."""
    from kvmini.engine.engine import EngineConfig, LLMEngine
    from kvmini.engine.scheduler import SamplingParams
    from kvmini.models.llama import get_config, random_weights
    from kvmini.models.reference import dense_greedy, dense_logits

    cfg = get_config("tiny")
    w = random_weights(cfg, "cpu", torch.float32, seed=2)
    eng = LLMEngine(EngineConfig(model="tiny", device="cpu", dtype="fp32", kernels="torch", max_model_len=128, max_num_seqs=4, max_batched_tokens=32, use_graphs=False,
                                 async_scheduling=async_scheduling), weights=w)
    prompt = [1, 7, 9, 30, 2, 44]
    plain = dense_greedy(w, prompt, 12)

    def biased(bias, banned_until=0, banned=()):
        ids, out = list(prompt), []
        for i in range(12):
            lg = dense_logits(w, ids)[-1].clone()
            for t, v in bias.items():
                lg[t] += v
            if i < banned_until:
                lg[list(banned)] = float("-inf")
            out.append(int(lg.argmax()))
            ids.append(out[-1])
        return out

    # a ban on the tokens the plain run picks, and a push for a token it never does
    bias = {plain[0]: -100.0, plain[1]: -100.0, 77: 6.0}
    got = eng.generate([prompt, prompt], [SamplingParams(max_tokens=12, ignore_eos=True, logit_bias=bias), SamplingParams(max_tokens=12, ignore_eos=True)])
    assert got[0] == biased(bias) and got[1] == plain
    assert plain[0] not in got[0] and plain[1] not in got[0]
    # min_tokens: a token of the plain run as a stop token ends the reply there; with min_tokens = 8 it cannot be picked before
    k = next(i for i in range(1, 8) if plain[i] not in plain[:i])
    stop = plain[k]
    short = eng.generate([prompt], SamplingParams(max_tokens=12, ignore_eos=True, stop_token_ids=[stop]))[0]
    assert short == plain[: k + 1]
    held = eng.generate([prompt], SamplingParams(max_tokens=12, stop_token_ids=[stop], min_tokens=8))[0]
    want = biased({}, banned_until=8, banned=(stop, cfg.eos_token_id))
    cut = next((i for i, t in enumerate(want) if i >= 8 and t in (stop, cfg.eos_token_id)), len(want) - 1)
    assert held == want[: cut + 1] and len(held) >= 8 and stop not in held[:8]
    # refused at intake: ids outside the vocabulary (they would index past the logits), values out of range, min > max
    for bad, word in ((dict(logit_bias={cfg.vocab_size: 1.0}), "logit_bias"), (dict(logit_bias={3: 1e9}), "logit_bias"), (dict(logit_bias={-1: 1.0}), "logit_bias"),
                      (dict(min_tokens=99), "min_tokens"), (dict(min_tokens=2, stop_token_ids=[cfg.vocab_size + 5]), "min_tokens")):
        seq = eng.make_sequence(prompt, SamplingParams(max_tokens=12, **bad))
        assert word in (eng.scheduler.validate(seq) or ""), bad
