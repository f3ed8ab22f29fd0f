"""Lock-step comparison of two engines over one greedy request (test infrastructure).

The product engine (HIP runners on the MI355X) and the oracle engine (CPU restatement of the reference, oracle/runner.py)
are driven ONE SPECULATION ROUND AT A TIME over the same request; after every round the harness compares what the draft
proposed (cache-hit flag, the K speculated tokens), what the target accepted (the suffix, token by token) and therefore
the accepted length.  Reference behaviour under test: ssd/engine/step.py:91-163 (SpecDecodeStep.decode),
ssd/engine/draft_runner.py:186-378 (hit_cache_and_respond + tree round), ssd/engine/speculator_sync.py:25-69.

Two bf16 pipelines that accumulate in different orders legitimately flip NEAR-TIES.  A difference is excused only when
the ORACLE's own margin at exactly that decision is a near-tie:
  * a target decision (accepted token j of a round)  -> the oracle target's top-2 margin at that position
    (OracleRunner.margin_log);
  * a speculated token after a cache hit             -> the oracle draft's top-2 gaps of the tree branches forked from the
    glue row the request named (decision_gaps "tree");
  * a speculated token after a miss (JIT chain) or in the synchronous chain -> that chain step's gap ("jit" / "chain");
  * a hit flag                                       -> the fork gap of the glue row the request named ("glue").
After an excused difference the two runs no longer see the same inputs, so both are RE-SYNCHRONISED: aborted and
restarted from prompt + the oracle's tokens up to and including the disputed one (teacher forcing), and the comparison
goes on.  (Round 3 stopped comparing at the first near-tie: 16 of 40 tokens.)  Anything not excused raises.
"""
from __future__ import annotations

from dataclasses import dataclass, field

NEAR_TIE = 0.0625           # one bf16 ulp at logit magnitude 8..16 (tests/util.py)


def _as_list(x):
    if x is None:
        return None
    return x if isinstance(x, list) else x.tolist()


class Traced:
    """One engine, stepped by hand (LLMEngine.add_request / step / abort_all), recording every speculation round."""

    def __init__(self, eng):
        self.eng = eng
        self.rounds: list[dict] = []
        self.step = None

    def start(self, prompt, sp) -> None:
        eng = self.eng
        eng.add_request(list(prompt), sp)
        self.step = eng.create_inference_step(eng.config)
        self.rounds = []
        rec = self.rounds
        spec_fn, ver_fn = self.step.speculator.speculate, self.step.verifier.verify

        def speculate(seqs, vr):
            out = spec_fn(seqs, vr)
            rec.append({"hit": None if out.cache_hits is None else int(_as_list(out.cache_hits)[0])})
            return out

        def verify(seqs, spec, eagle=False):
            out = ver_fn(seqs, spec, eagle=eagle)
            rec[-1]["spec"] = [int(t) for t in _as_list(spec.speculations[0])]       # [recovery, x_1 .. x_K] (device tensor: read after the verify)
            rec[-1]["accepted"] = [int(t) for t in out.new_suffixes[0]]
            return out
        self.step.speculator.speculate, self.step.verifier.verify = speculate, verify

    def advance(self) -> dict | None:
        """Run engine steps until one more speculation round is recorded (prefill steps pass through); None when finished."""
        n = len(self.rounds)
        while not self.eng.is_finished():
            self.eng.step(self.step)
            if len(self.rounds) > n:
                return self.rounds[-1]
        return None

    def abort(self) -> None:
        self.eng.abort_all()
        self.step = None


@dataclass
class Report:
    tokens: int = 0                 # stream tokens the request produced (teacher-forced ones included)
    tokens_compared: int = 0        # ... compared and found equal
    rounds: int = 0                 # oracle speculation rounds
    rounds_compared: int = 0        # ... whose hit flag, K speculated tokens and accepted suffix were all compared
    hits: int = 0
    misses: int = 0
    real_misses: int = 0            # ... not counting the first request of a (re)started run, which always misses
    partial_accepts: int = 0        # rounds with 1 < accepted length < K + 1
    full_accepts: int = 0
    restarts: int = 0
    excused: list = field(default_factory=list)
    accepted_lens: list = field(default_factory=list)

    def summary(self) -> str:
        return (f"{self.tokens_compared}/{self.tokens} tokens and {self.rounds_compared}/{self.rounds} rounds compared, "
                f"{self.restarts} re-synchronisations, hits {self.hits} misses {self.misses} (beyond first requests: {self.real_misses}) partial {self.partial_accepts} "
                f"full {self.full_accepts}; accepted lens {self.accepted_lens}; excused {self.excused}")


def _gaps_since(runner, start):
    out = {}
    for kind, g in runner.decision_gaps[start:]:
        out[kind] = g           # at most one of each kind per round
    return out


def compare_lockstep(prod_eng, oracle_eng, prompt, n_new: int, make_sp, fan_out: int | None = None, thr: float = NEAR_TIE,
                     max_restarts: int = 16, what: str = "") -> Report:
    """Drive both engines over `prompt` for n_new greedy tokens.  make_sp(max_new_tokens) -> SamplingParams.
    fan_out: uniform async fan-out F (None for synchronous speculation)."""
    K = prod_eng.config.speculate_k
    rep = Report()
    draft = getattr(oracle_eng, "draft_runner", None)
    target = oracle_eng.model_runner
    if draft is not None:
        draft.log_decisions = True
    forced: list[int] = []          # the oracle's stream so far (what both engines are teacher-forced with at a restart)
    while len(forced) < n_new:
        P, O = Traced(prod_eng), Traced(oracle_eng)
        cur_prompt = list(prompt) + forced
        sp = make_sp(n_new - len(forced))
        P.start(cur_prompt, sp)
        O.start(cur_prompt, sp)
        target.margin_log.clear()
        prev_gaps, prev_acc = {}, None
        diverged = False
        while len(forced) < n_new:
            g0 = len(draft.decision_gaps) if draft is not None else 0
            w = O.advance()
            g = P.advance()
            assert (w is None) == (g is None), f"{what}: one engine finished before the other"
            if w is None:
                break
            gaps = _gaps_since(draft, g0) if draft is not None else {}
            rep.rounds += 1
            rep.accepted_lens.append(len(w["accepted"]))
            rep.hits += w["hit"] == 1
            rep.misses += w["hit"] == 0
            rep.real_misses += w["hit"] == 0 and len(O.rounds) > 1
            rep.partial_accepts += 1 < len(w["accepted"]) < K + 1
            rep.full_accepts += len(w["accepted"]) == K + 1
            base = len(cur_prompt) - len(prompt) + sum(len(r["accepted"]) for r in O.rounds[:-1])       # completion tokens before this round
            room = n_new - base                                                                        # tokens of this round that still count
            excuse = None
            # ---- what the draft proposed ----
            if g["spec"][0] != w["spec"][0]:
                pass        # the recovery token (decided by the prefill / the previous verify) differs: accepted[0] below, a target decision
            elif g["hit"] != w["hit"]:
                row = (prev_acc or 1) - 1
                gap = float(prev_gaps["glue"][0, row]) if "glue" in prev_gaps else None
                assert gap is not None and gap <= thr, f"{what}: round {rep.rounds}: hit flag {g['hit']} vs oracle {w['hit']}, fork gap of glue row {row} = {gap}"
                excuse = ("hit", rep.rounds, gap)
            elif g["spec"] != w["spec"]:
                j = next(i for i in range(1, K + 1) if g["spec"][i] != w["spec"][i]) - 1            # chain / tree depth of the first differing token
                if w["hit"] == 1:
                    row = prev_acc - 1
                    tg = prev_gaps["tree"][: j + 1, row * fan_out:(row + 1) * fan_out]
                    gap = float(tg.min())
                    kind = "tree"
                else:
                    kind = "chain" if w["hit"] is None else "jit"
                    gap = float(gaps[kind][j, 0])
                assert gap <= thr, f"{what}: round {rep.rounds}: speculated token {j} differs ({g['spec']} vs {w['spec']}), oracle {kind} gap {gap}"
                excuse = (kind, rep.rounds, j, gap)
            # ---- what the target accepted (same target stream whatever the draft proposed) ----
            ga, wa = g["accepted"][:room], w["accepted"][:room]
            n_same = 0
            while n_same < min(len(ga), len(wa)) and ga[n_same] == wa[n_same]:
                n_same += 1
            flipped = n_same < min(len(ga), len(wa))
            if flipped:
                pos = len(prompt) + base + n_same
                m = next((v for (sid, p), v in target.margin_log.items() if p == pos), None)
                assert m is not None and m <= thr, (f"{what}: round {rep.rounds}: accepted token {n_same} differs ({ga} vs {wa}) although the oracle "
                                                    f"target's margin at position {pos} is {m}")
                excuse = ("target", rep.rounds, n_same, m)
            elif excuse is None and len(ga) != len(wa):
                # same speculation, one run accepted a draft token the other rejected: the target's decision right behind the
                # shorter suffix flipped (it shows up as the NEXT round's recovery token)
                pos = len(prompt) + base + n_same
                m = next((v for (sid, p), v in target.margin_log.items() if p == pos), None)
                assert m is not None and m <= thr, (f"{what}: round {rep.rounds}: same speculation {w['spec']}, accepted {ga} vs oracle {wa}; the "
                                                    f"oracle target's margin at position {pos} is {m}")
                excuse = ("target-len", rep.rounds, n_same, m)
            rep.tokens_compared += n_same
            if excuse is None:
                rep.rounds_compared += 1
                forced.extend(wa)
                prev_gaps, prev_acc = gaps, len(w["accepted"])
                continue
            # ---- excused near-tie: re-synchronise both runs on the oracle's tokens ----
            rep.excused.append(excuse)
            # (a draft-side excuse: go on from the tokens BOTH runs committed in this round -- at least the recovery token)
            forced.extend(wa[: n_same + 1] if flipped else wa[:n_same])
            diverged = True
            break
        P.abort()
        O.abort()
        if diverged:
            rep.restarts += 1
            assert rep.restarts <= max_restarts, f"{what}: more than {max_restarts} re-synchronisations: {rep.summary()}"
    rep.tokens = len(forced)
    return rep
