"""A training step replayed as ONE HIP graph (ssecg/graph.py) against the same steps run eagerly: bit-identical parameters,
optimiser state, BatchNorm buffers and logged statistics - with a learning rate that changes every step, real dropout seeds
from torch's CPU generator and fresh batches."""
import copy

import pytest
import torch

from helpers import TRAIN_CFG, build_hip_model
from ssecg import synth

pytestmark = pytest.mark.gpu


def _batches(n, B, C, L, dev):
    out = []
    for i in range(n):
        b = synth.fixmatch_batch(100 + i, B, C, L)
        out.append((torch.from_numpy(b["labeled"]["ecg"]).to(dev), torch.from_numpy(b["labeled"]["target"]).to(dev),
                    torch.from_numpy(b["unlabeled"]["ecg"]).to(dev), torch.from_numpy(b["unlabeled"]["ecg_aug"]).to(dev)))
    return out


def _run(dev, amp, graph, nsteps, B=4, C=2, L=500):
    import utils.lr_sched as lr_sched
    from algorithms.fixmatch import fixmatch_step
    from ssecg.graph import StepGraph
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    model = build_hip_model(C, synth.model_state(5, C, trained=True), dev)
    if amp:
        from ssecg import amp as SAMP
        SAMP.enable(model)
    cfg = dict(TRAIN_CFG)
    opt = get_optimizer_from_config(cfg, model.parameters())
    scaler = NativeScalerWithGradNormCount()
    torch.manual_seed(1234)                    # the dropout seeds come from this generator, eager and replayed alike

    def whole_step(ecg_x, mask_x, ecg_u_w, ecg_u_s):
        loss, stats = fixmatch_step(model, ecg_x, mask_x, ecg_u_w, ecg_u_s, 0.3)
        scaler(loss, opt, clip_grad=None, parameters=model.parameters(), update_grad=True)
        opt.zero_grad()
        return stats

    step = StepGraph(whole_step) if graph else whole_step
    stats = []
    for i, b in enumerate(_batches(nsteps, B, C, L, dev)):
        lr_sched.adjust_learning_rate(opt, 3.0 + i / 7.0, cfg)     # warm-up part of the schedule: a new lr every step
        stats.append(step(*b).clone())
    torch.cuda.synchronize()
    if graph:
        assert step.graph is not None and step.replays == nsteps - 2, step.replays
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    osd = copy.deepcopy(opt.state_dict())
    return sd, osd, torch.stack(stats).cpu(), scaler.state_dict()


@pytest.mark.parametrize("amp", [False, True], ids=["fp32", "bf16"])
def test_graph_replay_is_bit_identical_to_eager_steps(amp, dev):
    n = 7
    sd_e, osd_e, st_e, sc_e = _run(dev, amp, False, n)
    sd_g, osd_g, st_g, sc_g = _run(dev, amp, True, n)
    assert torch.equal(st_e, st_g), (st_e - st_g).abs().max()
    assert st_e[:, 3].min() > 0.0                      # the masked term is live (mask_ratio > 0)
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_g[k]), k
    for pe, pg in zip(osd_e["state"].values(), osd_g["state"].values()):
        assert float(pe["step"]) == float(pg["step"]) == n
        assert torch.equal(pe["exp_avg"], pg["exp_avg"]) and torch.equal(pe["exp_avg_sq"], pg["exp_avg_sq"])
    assert sc_e == sc_g
    assert len({float(v) for v in st_e[:, 0]}) == n    # seven different losses: the batches and the weights did change


def test_graph_replay_with_a_skipped_step_matches_eager(dev):
    """A non-finite batch inside the replayed region: GradScaler's inf-skip happens on the device (the optimiser kernel returns
    early, the skip counter and the scale move) and the host bookkeeping of the replay (per-parameter ``step``, reconciled at
    checkpoint time) must agree with the eager run's - weights, moments, step counts and scaler state."""
    n, bad = 8, 4

    def run(graph):
        import utils.lr_sched as lr_sched
        from algorithms.fixmatch import fixmatch_step
        from ssecg.graph import StepGraph
        from utils.misc import NativeScalerWithGradNormCount
        from utils.optimizer import get_optimizer_from_config
        model = build_hip_model(2, synth.model_state(5, 2, trained=True), dev)
        cfg = dict(TRAIN_CFG)
        opt = get_optimizer_from_config(cfg, model.parameters())
        scaler = NativeScalerWithGradNormCount()
        torch.manual_seed(4321)

        def whole_step(ecg_x, mask_x, ecg_u_w, ecg_u_s):
            loss, stats = fixmatch_step(model, ecg_x, mask_x, ecg_u_w, ecg_u_s, 0.3)
            scaler(loss, opt, clip_grad=None, parameters=model.parameters(), update_grad=True)
            opt.zero_grad()
            return stats

        step = StepGraph(whole_step) if graph else whole_step
        for i, b in enumerate(_batches(n, 4, 2, 500, dev)):
            lr_sched.adjust_learning_rate(opt, 3.0 + i / 7.0, cfg)
            if i == bad:
                b = (b[0].clone(), b[1], b[2], b[3])
                b[0][0, 0, 10] = float("inf")           # one labelled sample is non-finite: this step must be skipped
            step(*b)
        torch.cuda.synchronize()
        return ({k: v.detach().clone() for k, v in model.state_dict().items()}, copy.deepcopy(opt.state_dict()), scaler.state_dict())

    sd_e, osd_e, sc_e = run(False)
    sd_g, osd_g, sc_g = run(True)
    assert sc_e == sc_g and sc_e["scale"] == 65536.0 * 0.5            # one backoff
    for k in sd_e:
        if sd_e[k].dtype.is_floating_point and "running" not in k:
            assert torch.isfinite(sd_e[k]).all(), k                     # the skipped step left the weights alone
        assert torch.equal(sd_e[k], sd_g[k]) or (torch.isnan(sd_e[k]) == torch.isnan(sd_g[k])).all(), k
    for pe, pg in zip(osd_e["state"].values(), osd_g["state"].values()):
        assert float(pe["step"]) == float(pg["step"]) == n - 1          # the skipped launch is no optimiser step
        assert torch.equal(pe["exp_avg"], pg["exp_avg"]) and torch.equal(pe["exp_avg_sq"], pg["exp_avg_sq"])


def test_graph_falls_back_to_eager_on_another_shape(dev):
    from ssecg.graph import StepGraph
    calls = []

    def fn(x):
        calls.append(tuple(x.shape))
        return (x * 2.0,)

    g = StepGraph(fn)
    a = torch.ones(4, 8, device=dev)
    for _ in range(4):
        assert torch.equal(g(a)[0], a * 2.0)
    assert g.replays == 2 and len(calls) == 3          # two eager calls + the capture pass
    b = torch.ones(2, 8, device=dev)
    assert torch.equal(g(b)[0], b * 2.0) and calls[-1] == (2, 8) and g.replays == 2


def _fixmatch_runner(dev, seed, C=2, graph=True):
    """-> (model, optimiser, step callable, scaler): one FixMatch trainer (its own model / AdamW / GradScaler)."""
    from algorithms.fixmatch import fixmatch_step
    from ssecg.graph import StepGraph
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    model = build_hip_model(C, synth.model_state(seed, C, trained=True), dev)
    cfg = dict(TRAIN_CFG)
    opt = get_optimizer_from_config(cfg, model.parameters())
    scaler = NativeScalerWithGradNormCount()

    def whole_step(ecg_x, mask_x, ecg_u_w, ecg_u_s):
        loss, stats = fixmatch_step(model, ecg_x, mask_x, ecg_u_w, ecg_u_s, 0.3)
        scaler(loss, opt, clip_grad=None, parameters=model.parameters(), update_grad=True)
        opt.zero_grad()
        return stats

    return model, opt, (StepGraph(whole_step) if graph else whole_step), scaler, cfg


def _state(model, opt):
    return ({k: v.detach().clone() for k, v in model.state_dict().items()}, copy.deepcopy(opt.state_dict()))


def _assert_same(a, b, nsteps):
    (sd_a, osd_a), (sd_b, osd_b) = a, b
    for k in sd_a:
        assert torch.equal(sd_a[k], sd_b[k]), k
    for pa, pb in zip(osd_a["state"].values(), osd_b["state"].values()):
        assert float(pa["step"]) == float(pb["step"]) == nsteps
        assert torch.equal(pa["exp_avg"], pb["exp_avg"]) and torch.equal(pa["exp_avg_sq"], pb["exp_avg_sq"])


def test_replay_survives_a_freed_model_that_the_capture_saw(dev):
    """the refresh launches a step graph captures cover EVERY weight registered in the process-global operand
    caches - also those of a model the step does not use (the ST++ stage-1 model while stage 2 captures).  Model A trains with
    its own graph; model B's graph is captured while A is alive; A and its graph are freed, an eval pass refreshes the caches,
    new tensors are allocated into the freed blocks - and B's replays must still equal B's all-eager run bit for bit (the graph
    owns what its captured launches touch: ssecg/graph.py, Ownership)."""
    import gc
    import utils.lr_sched as lr_sched
    batches = _batches(9, 4, 2, 500, dev)

    def run_b(graph, with_a):
        torch.manual_seed(99)
        if with_a:
            mA, oA, stepA, _, cfgA = _fixmatch_runner(dev, 7, graph=True)
            for i in range(4):
                lr_sched.adjust_learning_rate(oA, 3.0 + i / 7.0, cfgA)
                stepA(*batches[i])
            assert stepA.graph is not None
            torch.manual_seed(99)                    # B's dropout seeds: the same stream as in the run without A
        mB, oB, stepB, _, cfgB = _fixmatch_runner(dev, 5, graph=graph)
        for i in range(4):
            lr_sched.adjust_learning_rate(oB, 3.0 + i / 7.0, cfgB)
            stepB(*batches[i])
        if with_a:
            assert stepB.graph is not None and stepB.replays == 2
            stepA.release()
            del mA, oA, stepA
            gc.collect(); torch.cuda.empty_cache()
            with torch.no_grad():                   # an eager eval pass: the caches drop what died and re-make their tables
                mB.eval(); mB(batches[0][2], return_loss=False); mB.train()
            junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(64)]   # recycle whatever was freed
        for i in range(4, 9):
            lr_sched.adjust_learning_rate(oB, 3.0 + i / 7.0, cfgB)
            stepB(*batches[i])
        torch.cuda.synchronize()
        return _state(mB, oB)

    _assert_same(run_b(False, False), run_b(True, True), 9)


def test_short_batch_between_replays_runs_eagerly_on_the_same_state(dev):
    """capture, a SHORTER batch (eager fallback on the state the graph also owns: parameters, moments, gradient
    tensors, pointer tables), then more replays - against the all-eager run, bit for bit (weights, moments, step counts)."""
    import utils.lr_sched as lr_sched
    full = _batches(8, 4, 2, 500, dev)
    short = tuple(t[:2].clone() for t in _batches(1, 4, 2, 500, dev)[0])
    seq = full[:5] + [short] + full[5:]

    def run(graph):
        torch.manual_seed(77)
        m, o, step, _, cfg = _fixmatch_runner(dev, 5, graph=graph)
        for i, b in enumerate(seq):
            lr_sched.adjust_learning_rate(o, 3.0 + i / 7.0, cfg)
            step(*b)
        torch.cuda.synchronize()
        if graph:
            assert step.graph is not None and step.replays == len(seq) - 3   # two warm-up steps and the short batch ran eagerly
        return _state(m, o)

    _assert_same(run(False), run(True), len(seq))


def test_a_failed_capture_falls_back_to_eager_with_the_host_state_rolled_back(dev, monkeypatch):
    """a capture that raises (here: more per-step scalars than the block holds) must not kill the run nor leave the
    host bookkeeping advanced (per-parameter ``step``, the dropout seed drawn from torch's CPU generator): the StepGraph runs
    eagerly from then on and the run equals the all-eager one bit for bit."""
    import utils.lr_sched as lr_sched
    from ssecg import graph as G
    batches = _batches(6, 4, 2, 500, dev)

    def run(graph):
        torch.manual_seed(55)
        m, o, step, _, cfg = _fixmatch_runner(dev, 5, graph=graph)
        for i, b in enumerate(batches):
            lr_sched.adjust_learning_rate(o, 3.0 + i / 7.0, cfg)
            step(*b)
        torch.cuda.synchronize()
        if graph:
            assert step.disabled and step.graph is None and step.replays == 0
        return _state(m, o)

    eager = run(False)
    monkeypatch.setattr(G, "_WORDS", 3)          # an AdamW group needs 5 words: the capture raises inside torch.cuda.graph
    _assert_same(eager, run(True), len(batches))


def test_a_failed_capture_drops_the_gradients_it_recorded(dev, monkeypatch):
    """the capture above raises inside optimizer.step, AFTER the captured backward - every ``.grad`` then points at
    capture-pool memory whose producing kernels never ran, and the eager re-run's AccumulateGrad would add into it (the
    bit-for-bit test above could only pass on zero-filled fresh VRAM).  The abort path must leave ``.grad`` None: probed
    at the entry of the eager re-run, not inferred from memory contents.  Same for a capture that dies inside the backward."""
    import utils.lr_sched as lr_sched
    from ssecg import graph as G
    from ssecg import ops
    batches = _batches(4, 4, 2, 500, dev)
    for where in ("optimizer", "backward"):
        torch.manual_seed(55)
        m, o, step, _, cfg = _fixmatch_runner(dev, 5, graph=True)
        inner, seen = step.step_fn, []

        def probed(*a, inner=inner, seen=seen, m=m):
            seen.append((ops.STEP_SCALARS is not None, all(p.grad is None for p in m.parameters())))
            return inner(*a)

        step.step_fn = probed
        if where == "optimizer":
            monkeypatch.setattr(G, "_WORDS", 3)
        else:
            monkeypatch.setattr(G, "_WORDS", 512)
            h = m.backbone.stem[0].weight.register_hook(lambda g: (_ for _ in ()).throw(RuntimeError("injected failure in the captured backward"))
                                                        if ops.STEP_SCALARS is not None else None)
        for i, b in enumerate(batches):
            lr_sched.adjust_learning_rate(o, 3.0 + i / 7.0, cfg)
            step(*b)
        torch.cuda.synchronize()
        if where == "backward":
            h.remove()
        assert step.disabled and step.replays == 0
        # calls: 2 eager warm-ups, the capture attempt (STEP_SCALARS set), its eager re-run, the remaining eager step
        assert [c for c, _ in seen] == [False, False, True, False, False], seen
        assert all(none for _, none in seen), f"{where}: a step started with stale gradients: {seen}"
        assert all(torch.isfinite(p).all() for p in m.parameters())


@pytest.mark.parametrize("algo", ["fixmatch", "mean_teacher", "base", "cps", "stpp"])
def test_plugin_epoch_with_hip_graph(algo, dev):
    """``train.hip_graph: true`` through the plugins' own epoch loops (loaders = lists of batch dicts): the meters and the
    weights (student, and MeanTeacher's EMA teacher) after the epoch equal the eager epoch's, and the graph really replayed."""
    import algorithms.base as A_base
    import algorithms.fixmatch as A_fm
    import algorithms.mean_teacher as A_mt
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    C, B, L, n = 2, 4, 500, 6
    batches = [synth.fixmatch_batch(300 + i, B, C, L) for i in range(n)]
    lab = [{k: torch.from_numpy(v) for k, v in b["labeled"].items()} for b in batches]
    unl = [{k: torch.from_numpy(v) for k, v in b["unlabeled"].items()} for b in batches]
    res = {}
    for mode in (False, True):
        model = build_hip_model(C, synth.model_state(6, C, trained=True), dev)
        cfg = dict(TRAIN_CFG, conf_thresh=0.3, hip_graph=mode)
        opt = get_optimizer_from_config(cfg, model.parameters())
        scaler = NativeScalerWithGradNormCount()
        torch.manual_seed(77)
        teacher = None
        if algo == "fixmatch":
            stats = A_fm.train_one_epoch(model, lab, unl, opt, dev, 2, scaler, None, use_amp=False, config=cfg)
        elif algo == "mean_teacher":
            teacher = build_hip_model(C, synth.model_state(6, C, trained=True), dev)
            for p in teacher.parameters():
                p.requires_grad = False
            with torch.no_grad():
                for pq, pk in zip(model.parameters(), teacher.parameters()):
                    pk.data = pq.data                   # src/algorithms/mean_teacher.py:285-290 (Q4): un-aliased by the first EMA
            stats = A_mt.train_one_epoch(model, teacher, lab, unl, opt, dev, 2, scaler, None, use_amp=False, config=cfg)
        elif algo == "stpp":
            import algorithms.stpp as A_stpp
            teacher = build_hip_model(C, synth.model_state(9, C, trained=True), dev)     # frozen teacher of the stage
            for p in teacher.parameters():
                p.requires_grad = False
            stats = A_stpp.train_one_epoch(model, teacher, lab, unl, opt, dev, 2, scaler, None, use_amp=False, config=cfg)
        elif algo == "cps":
            import algorithms.cps as A_cps
            teacher = build_hip_model(C, synth.model_state(9, C, trained=True), dev)     # the second model (both train)
            opt2 = get_optimizer_from_config(cfg, teacher.parameters())
            stats = A_cps.train_one_epoch(model, teacher, lab, unl, opt, opt2, dev, 2, scaler, None, use_amp=False, config=cfg)
        else:
            stats = A_base.train_one_epoch(model, lab, opt, dev, 2, scaler, None, use_amp=False, config=cfg)
        torch.cuda.synchronize()
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        if teacher is not None:
            sd.update({"teacher." + k: v.detach().clone() for k, v in teacher.state_dict().items()})
        res[mode] = (stats, sd)
        if mode:
            assert model._ssecg_step_graph.replays == n - 2
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    for k, v in res[False][1].items():
        assert torch.equal(v, res[True][1][k]), k


@pytest.mark.parametrize("amp", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hip_graph"])
def test_pass_overlap_is_bit_identical(amp, graph, dev, monkeypatch):
    """Small batches run the pseudo-label pass on a side HIP stream beside the student forward (ops.PassOverlap; graph edges
    under capture).  Same kernels in the same order per stream, shared operands formed before the fork: weights, optimiser
    state, BatchNorm buffers and per-step statistics must equal the single-stream run bit for bit, eagerly and replayed."""
    from ssecg import ops
    n = 7
    monkeypatch.setattr(ops, "OVERLAP_PASSES", "0")
    ref = _run(dev, amp, graph, n)
    monkeypatch.setattr(ops, "OVERLAP_PASSES", "1")
    forks = []
    orig = ops._TeacherSide.__enter__
    monkeypatch.setattr(ops._TeacherSide, "__enter__", lambda self: (forks.append(self.ov.on), orig(self))[1])
    got = _run(dev, amp, graph, n)
    assert any(forks), "the side stream was never used"
    (sd_a, osd_a, st_a, sc_a), (sd_b, osd_b, st_b, sc_b) = ref, got
    assert torch.equal(st_a, st_b)
    for k in sd_a:
        assert torch.equal(sd_a[k], sd_b[k]), k
    for pa, pb in zip(osd_a["state"].values(), osd_b["state"].values()):
        assert torch.equal(pa["exp_avg"], pb["exp_avg"]) and torch.equal(pa["exp_avg_sq"], pb["exp_avg_sq"])
    assert sc_a == sc_b


def test_pass_overlap_survives_an_exception_in_the_student_forward(dev, monkeypatch):
    """Round-5 verdict, weak #9: an exception between the fork and the join (an out-of-memory student forward, a KeyboardInterrupt)
    used to leave the process-wide "overlap active" slot set - every later step silently single-stream, and the main stream never
    ordered behind the side stream's kernels.  ``PassOverlap`` is a context manager now: leaving the block ALWAYS joins and clears
    the device's slot.  Steps 0-2 normal, step 3 raises inside the student forward, steps 3-6 re-run: the slot is empty after the
    exception, the next step forks again, and the whole trajectory equals an undisturbed run's bit for bit."""
    import utils.lr_sched as lr_sched
    from algorithms.fixmatch import fixmatch_step
    from ssecg import ops
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    if ops.OVERLAP_PASSES == "0":
        pytest.skip("SSECG_OVERLAP_PASSES=0")
    n, bad = 7, 3
    ref = _run(dev, False, False, n)
    model = build_hip_model(2, synth.model_state(5, 2, trained=True), dev)
    cfg = dict(TRAIN_CFG)
    opt = get_optimizer_from_config(cfg, model.parameters())
    scaler = NativeScalerWithGradNormCount()
    torch.manual_seed(1234)
    forks = []
    orig = ops._TeacherSide.__enter__
    monkeypatch.setattr(ops._TeacherSide, "__enter__", lambda self: (orig(self), forks.append(self.ov.on))[0])
    boom = {"armed": False}

    def hook(mod, inp):
        if boom["armed"] and mod.training:
            boom["armed"] = False
            raise RuntimeError("injected failure in the student forward")

    model.backbone.layer3.register_forward_pre_hook(hook)
    stats = []
    for i, b in enumerate(_batches(n, 4, 2, 500, dev)):
        lr_sched.adjust_learning_rate(opt, 3.0 + i / 7.0, cfg)
        if i == bad:
            snap = {k: v.detach().clone() for k, v in model.state_dict().items()}
            rng = torch.get_rng_state()
            boom["armed"] = True
            with pytest.raises(RuntimeError, match="injected failure"):
                fixmatch_step(model, *b, 0.3)
            assert not ops._overlap_active, "the overlap slot outlived the exception"
            assert ops._scope_depth[0] == 0
            from ssecg import functional as SF
            assert not SF._pending_counters, "num_batches_tracked increments of the failed forward were left pending"
            torch.cuda.synchronize()
            opt.zero_grad()
            torch.set_rng_state(rng)        # (the failed forward never reached the head: no dropout seed was drawn)
            # the failed train-mode forward updated the running statistics (and counters) of the BatchNorms it passed: a retry is a
            # new step for them.  Bit-identity with the undisturbed run needs them put back - what a checkpoint-based retry restores
            model.load_state_dict(snap)
        loss, st = fixmatch_step(model, *b, 0.3)
        scaler(loss, opt, clip_grad=None, parameters=model.parameters(), update_grad=True)
        opt.zero_grad()
        stats.append(st.clone())
    torch.cuda.synchronize()
    assert len(forks) == n + 1 and all(forks), forks        # every step - the failed one and the ones after it - ran on two streams
    assert torch.equal(torch.stack(stats).cpu(), ref[2])
    for k, v in model.state_dict().items():
        assert torch.equal(v, ref[0][k]), k


@pytest.mark.parametrize("amp", [False, True], ids=["fp32", "bf16"])
def test_two_models_alternating_with_overlap_and_hip_graph(amp, dev):
    """Round-5 verdict, weak #9: a second model's FIRST step after a first model's steps, with the side-stream pass on AND graph
    capture on - the operand caches (Winograd / bf16 weight operands, folded BatchNorm coefficients) are shared by every model of
    the process.  Two FixMatch models alternate steps in one process, each through its own StepGraph (model B's capture happens
    after model A's graph has replayed); each model's trajectory must equal the one it follows alone in a fresh run, bit for bit."""
    import utils.lr_sched as lr_sched
    from algorithms.fixmatch import fixmatch_step
    from ssecg import ops
    from ssecg.graph import StepGraph
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    if ops.OVERLAP_PASSES == "0":
        pytest.skip("SSECG_OVERLAP_PASSES=0")
    n = 6
    batches = _batches(n, 4, 2, 500, dev)

    def make(seed):
        model = build_hip_model(2, synth.model_state(seed, 2, trained=True), dev)
        if amp:
            from ssecg import amp as SAMP
            SAMP.enable(model)
        cfg = dict(TRAIN_CFG)
        opt = get_optimizer_from_config(cfg, model.parameters())
        scaler = NativeScalerWithGradNormCount()
        model.decode_head.dropout = None; model.decode_head.dropout_ratio = 0.0       # no host RNG: the two schedules draw nothing

        def whole_step(ecg_x, mask_x, ecg_u_w, ecg_u_s):
            loss, stats = fixmatch_step(model, ecg_x, mask_x, ecg_u_w, ecg_u_s, 0.3)
            scaler(loss, opt, clip_grad=None, parameters=model.parameters(), update_grad=True)
            opt.zero_grad()
            return stats

        return model, opt, cfg, StepGraph(whole_step)

    def alone(seed):
        model, opt, cfg, step = make(seed)
        out = []
        for i, b in enumerate(batches):
            lr_sched.adjust_learning_rate(opt, 3.0 + i / 7.0, cfg)
            out.append(step(*b).clone())
        torch.cuda.synchronize()
        assert step.graph is not None and step.replays == n - 2
        return torch.stack(out).cpu(), {k: v.detach().clone() for k, v in model.state_dict().items()}

    ref_a, ref_b = alone(5), alone(9)
    ma, oa, ca, sa = make(5)
    mb, ob, cb, sb = make(9)
    out_a, out_b = [], []
    # A runs three steps (two eager + its capture) before B's first step; then they alternate
    order = ["a", "a", "a", "b", "a", "b", "b", "a", "b", "a", "b", "b"]
    ia = ib = 0
    for who in order:
        if who == "a":
            lr_sched.adjust_learning_rate(oa, 3.0 + ia / 7.0, ca)
            out_a.append(sa(*batches[ia]).clone()); ia += 1
        else:
            lr_sched.adjust_learning_rate(ob, 3.0 + ib / 7.0, cb)
            out_b.append(sb(*batches[ib]).clone()); ib += 1
    torch.cuda.synchronize()
    assert ia == n and ib == n and sa.graph is not None and sb.graph is not None and sa.replays == n - 2 and sb.replays == n - 2
    assert not ops._overlap_active
    assert torch.equal(torch.stack(out_a).cpu(), ref_a[0]) and torch.equal(torch.stack(out_b).cpu(), ref_b[0])
    for k, v in ma.state_dict().items():
        assert torch.equal(v, ref_a[1][k]), "A " + k
    for k, v in mb.state_dict().items():
        assert torch.equal(v, ref_b[1][k]), "B " + k
