"""Lightweight quality evaluation + Pareto bucketing (``kvmini quality``).

A small embedded item bank — 3 commonsense-completion items (4-way multiple
choice), 3 passage yes/no items and 4 arithmetic items — is sent to the chat
endpoint at temperature 0; ``quality_score`` is the mean task accuracy × 100 and
``avg_quality_latency_ms`` the mean per-call latency, the keys the sweeps merge
into ``results.json`` (reference ``quality/evaluator.py:226-257``; the items here
are written for this build, the task mix and scoring rules are the same:
letter match, ``startswith('yes'|'no')``, digits-only equality).
``ParetoAnalyzer`` marks configurations on the (quality ↑, p95 ↓, $/1K tok ↓)
frontier (``:260-314``).
"""

from __future__ import annotations

import argparse
import json
import sys
import time
import urllib.error
import urllib.request
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from kvmini.utils import load_json, write_json

COMPLETION_ITEMS = [
    {"context": "A cook cracks two eggs into a hot pan and picks up a spatula. Next, the cook...", "choices": ["throws the pan out of the window.", "stirs the eggs as they begin to set.", "puts the spatula in the freezer.", "paints the kitchen wall."], "correct": 1},
    {"context": "A cyclist notices the rear tyre is flat and takes a pump from the bag. Then the cyclist...", "choices": ["attaches the pump to the valve and inflates the tyre.", "eats the pump.", "plants the pump in the garden.", "mails the tyre to a friend."], "correct": 0},
    {"context": "Rain starts falling while a man hangs laundry outside. He quickly...", "choices": ["adds more wet clothes to the line.", "sits down to sunbathe.", "starts washing the car.", "gathers the clothes and carries them indoors."], "correct": 3},
]
YESNO_ITEMS = [
    {"passage": "The Suez Canal links the Mediterranean Sea and the Red Sea. It opened in 1869.", "question": "Did the Suez Canal open in the nineteenth century?", "answer": True},
    {"passage": "The C language was developed by Dennis Ritchie in the early 1970s. Rust reached version 1.0 in 2015.", "question": "Is Rust older than C?", "answer": False},
    {"passage": "Lake Baikal is the deepest lake on Earth, reaching about 1,642 metres.", "question": "Is Lake Baikal deeper than 2,000 metres?", "answer": False},
]
MATH_ITEMS = [
    {"problem": "What is 19 + 23?", "answer": "42"},
    {"problem": "What is 9 × 6?", "answer": "54"},
    {"problem": "What is 132 ÷ 11?", "answer": "12"},
    {"problem": "What is 90 - 28?", "answer": "62"},
]

CallFn = Callable[[str, int], Tuple[str, float, int]]


class QualityEvaluator:
    DEFAULT_TASKS = ["hellaswag", "boolq", "arc_easy"]

    def __init__(self, endpoint: str, model_name: str, tasks: Optional[Sequence[str]] = None, num_samples: int = 100, api_key: Optional[str] = None, timeout: float = 30.0, call_fn: Optional[CallFn] = None):
        self.endpoint = endpoint.rstrip("/")
        self.model_name = model_name
        self.tasks = list(tasks or self.DEFAULT_TASKS)
        self.num_samples = num_samples
        self.api_key = api_key
        self.timeout = timeout
        self._call = call_fn or self._call_model

    def _call_model(self, prompt: str, max_tokens: int = 32) -> Tuple[str, float, int]:
        body = {"model": self.model_name, "messages": [{"role": "user", "content": prompt}], "max_tokens": max_tokens, "temperature": 0.0, "stream": False}
        headers = {"Content-Type": "application/json"}
        if self.api_key:
            headers["Authorization"] = f"Bearer {self.api_key}"
        req = urllib.request.Request(self.endpoint + "/v1/chat/completions", data=json.dumps(body).encode(), headers=headers)
        t0 = time.perf_counter()
        try:
            with urllib.request.urlopen(req, timeout=self.timeout) as r:
                data = json.loads(r.read())
        except (urllib.error.URLError, OSError, ValueError):
            return "", time.perf_counter() - t0, 0
        dt = time.perf_counter() - t0
        try:
            text = data["choices"][0]["message"]["content"] or ""
        except (KeyError, IndexError, TypeError):
            text = ""
        return text, dt, int((data.get("usage") or {}).get("completion_tokens") or 0)

    def _run(self, task: str, items: List[Dict[str, Any]], prompt_of: Callable[[Dict[str, Any]], str], max_tokens: int, judge: Callable[[str, Dict[str, Any]], bool]) -> Dict[str, Any]:
        items = items[: max(1, self.num_samples)]
        correct, lat = 0, 0.0
        for it in items:
            text, dt, _ = self._call(prompt_of(it), max_tokens)
            lat += dt
            correct += bool(judge(text, it))
        return {"task": task, "score": correct / len(items), "samples": len(items), "avg_latency": lat / len(items)}

    def _evaluate_hellaswag(self) -> Dict[str, Any]:
        def prompt(it: Dict[str, Any]) -> str:
            opts = "\n".join(f"{chr(65 + i)}) {c}" for i, c in enumerate(it["choices"]))
            return f"Complete this scenario by choosing the most likely next action:\n\n{it['context']}\n\nChoices:\n{opts}\n\nAnswer with just the letter (A, B, C, or D):"

        def judge(text: str, it: Dict[str, Any]) -> bool:
            a = text.strip().upper()
            return a in ("A", "B", "C", "D") and ord(a) - 65 == it["correct"]

        return self._run("hellaswag", COMPLETION_ITEMS, prompt, 1, judge)

    def _evaluate_boolq(self) -> Dict[str, Any]:
        def prompt(it: Dict[str, Any]) -> str:
            return f"Read the passage and answer the yes/no question:\n\nPassage: {it['passage']}\n\nQuestion: {it['question']}\n\nAnswer (Yes or No):"

        return self._run("boolq", YESNO_ITEMS, prompt, 1, lambda text, it: text.strip().lower().startswith("yes" if it["answer"] else "no"))

    def _evaluate_math(self) -> Dict[str, Any]:
        return self._run("math", MATH_ITEMS, lambda it: f"Solve this math problem and give just the number as your answer:\n{it['problem']}", 8, lambda text, it: "".join(c for c in text if c.isdigit()) == it["answer"])

    def evaluate(self) -> Dict[str, Any]:
        res: Dict[str, Any] = {}
        if "hellaswag" in self.tasks:
            res["hellaswag"] = self._evaluate_hellaswag()
        if "boolq" in self.tasks:
            res["boolq"] = self._evaluate_boolq()
        if "math" in self.tasks or "arc_easy" in self.tasks:
            res["math"] = self._evaluate_math()
        scores = [r["score"] for r in res.values()]
        lats = [r["avg_latency"] for r in res.values()]
        return {
            "quality_score": round(float(np.mean(scores)) * 100, 2) if scores else 0,
            "avg_quality_latency_ms": round(float(np.mean(lats)) * 1000, 1) if lats else 0,
            "task_results": res,
            "evaluated_at": time.strftime("%Y-%m-%dT%H:%M:%S"),
            "model_endpoint": self.endpoint,
            "model_name": self.model_name,
        }


AGREEMENT_TOPICS = ("the harbour at dawn", "a recipe for flat bread", "how rivers shape valleys", "the rules of a card game", "repairing a bicycle chain", "a letter to a landlord",
                    "the history of a small town", "why the sky is blue", "a train timetable", "packing for a week in the mountains", "a job interview", "the inventory of a hardware shop")
AGREEMENT_FORMS = ("Write two sentences about {t}.", "List three facts about {t}.", "Explain {t} to a child.", "Summarise what you know about {t} in one paragraph.",
                   "What is surprising about {t}?", "Give a short title for an essay on {t}.", "Ask a question about {t}.", "Describe {t} in five words.")


def agreement_prompts(n: int) -> List[str]:
    """``n`` distinct fixed prompts from the topic x form grid (96 cells), forms and topics both
    changing from one prompt to the next."""
    nf, nt = len(AGREEMENT_FORMS), len(AGREEMENT_TOPICS)
    return [AGREEMENT_FORMS[i % nf].format(t=AGREEMENT_TOPICS[(i // nf + i % nf) % nt]) for i in range(min(n, nf * nt))]


def prefix_agreement(reference: Sequence[str], got: Sequence[str]) -> float:
    """Mean over prompts of (words of the greedy continuation that match the reference's before
    the first difference) / (words of the reference).  Greedy decoding is a deterministic chain, so
    the first difference is where the two models' argmax parted; what follows is conditioned on
    different text and says nothing more."""
    if not reference:
        return float("nan")
    total = 0.0
    for ref, out in zip(reference, got):
        a, b = ref.split(), out.split()
        if not a:
            total += 1.0 if not b else 0.0
            continue
        k = 0
        while k < len(a) and k < len(b) and a[k] == b[k]:
            k += 1
        total += k / len(a)
    return total / len(reference)


def first_word_agreement(reference: Sequence[str], got: Sequence[str]) -> float:
    """Share of prompts whose greedy continuation starts with the reference's first word: both
    models saw exactly the same context there, so this is their top-1 agreement."""
    if not reference:
        return float("nan")
    same = sum(1 for ref, out in zip(reference, got) if ref.split()[:1] == out.split()[:1])
    return same / len(reference)


class AgreementEvaluator:
    """Fidelity of a quantised configuration measured against a reference configuration of the
    SAME model on greedy continuations of fixed prompts.  ``quality_score`` = 100 x
    ``first_word_agreement`` (top-1 agreement on identical contexts); ``prefix_agreement`` (how far
    the two deterministic chains run together) is reported next to it — with random-init weights
    the logits are nearly flat and a chain parts at its first flipped argmax, so the prefix score
    collapses towards 0 for every quantised format while the top-1 score still ranks them.

    The task-accuracy score of :class:`QualityEvaluator` needs trained weights; with random-init
    weights (no checkpoint on the machine) it is noise, while agreement with the unquantised
    engine still ranks weight and KV formats by how much they disturb the logits — the quality
    axis of the quantisation sweep's Pareto analysis (reference ``sweeps/quantization_sweep.py:286-310``
    merges ``quality_score`` into each row the same way)."""

    def __init__(self, endpoint: str, model_name: str, num_prompts: int = 96, max_tokens: int = 8, timeout: float = 60.0, call_fn: Optional[CallFn] = None):
        self.inner = QualityEvaluator(endpoint, model_name, [], 0, timeout=timeout, call_fn=call_fn)
        self.prompts = agreement_prompts(num_prompts)
        self.max_tokens = max_tokens

    def continuations(self) -> Tuple[List[str], float]:
        texts, lat = [], 0.0
        for p in self.prompts:
            text, dt, _ = self.inner._call(p, self.max_tokens)
            texts.append(text)
            lat += dt
        return texts, lat / max(1, len(self.prompts))

    def evaluate(self, reference: Optional[Sequence[str]]) -> Dict[str, Any]:
        """``reference`` None: this IS the reference configuration (score 100, continuations returned
        for the others to be compared with)."""
        texts, lat = self.continuations()
        if not any(texts):
            raise RuntimeError("the endpoint returned no text for any agreement prompt")
        score = 100.0 if reference is None else round(100.0 * first_word_agreement(reference, texts), 2)
        prefix = 100.0 if reference is None else round(100.0 * prefix_agreement(reference, texts), 2)
        return {"quality_score": score, "prefix_agreement": prefix, "avg_quality_latency_ms": round(lat * 1000, 1), "quality_metric": "greedy_top1_agreement", "continuations": texts}


class ParetoAnalyzer:
    """3-D frontier on (-quality, p95_ms, cost_per_1k_tokens): a point is
    dominated when another is <= in all three and < in at least one."""

    @staticmethod
    def frontier_mask(points: np.ndarray) -> np.ndarray:
        """``points[n, d]`` with lower-is-better axes -> bool mask of non-dominated rows."""
        p = np.asarray(points, dtype=np.float64)
        if p.size == 0:
            return np.zeros(0, dtype=bool)
        if p.shape[0] >= 64:  # the broadcast below is n*n*d doubles; the C++ loop exits early instead
            from kvmini.analysis import native_host

            if native_host.available():
                return native_host.pareto_mask(p)
        le = (p[None, :, :] <= p[:, None, :]).all(-1)  # le[i, j]: j <= i everywhere
        lt = (p[None, :, :] < p[:, None, :]).any(-1)
        dominated = (le & lt).any(1)
        return ~dominated

    @staticmethod
    def classify_pareto_bucket(results: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        if not results:
            return results
        inf = float("inf")

        def num(v: Any, default: float) -> float:
            try:
                f = float(v)
                return f if f == f else default
            except (TypeError, ValueError):
                return default

        pts = np.array([[-num(r.get("quality_score"), 0.0), num(r.get("p95_ms"), inf), num(r.get("cost_per_1k_tokens"), inf)] for r in results])
        mask = ParetoAnalyzer.frontier_mask(pts)
        return [{**r, "pareto_bucket": "on-pareto" if m else "off-pareto"} for r, m in zip(results, mask)]


def integrate_quality_eval(results_file: str, endpoint: str, model: str, **kw: Any) -> Dict[str, Any]:
    results = load_json(results_file)
    q = QualityEvaluator(endpoint, model, **kw).evaluate()
    results.update(q)
    write_json(results_file, results)
    return q


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="kvmini quality", description="LLM quality evaluation")
    ap.add_argument("--endpoint", required=True)
    ap.add_argument("--model", required=True)
    ap.add_argument("--results-file", default=None, help="results.json to augment in place")
    ap.add_argument("--tasks", nargs="+", default=QualityEvaluator.DEFAULT_TASKS)
    ap.add_argument("--samples", type=int, default=100)
    ap.add_argument("--api-key", default=None)
    ap.add_argument("--output", default=None)
    a = ap.parse_args(argv)
    if a.results_file:
        q = integrate_quality_eval(a.results_file, a.endpoint, a.model, tasks=a.tasks, num_samples=a.samples, api_key=a.api_key)
        print(f"Quality evaluation integrated into {a.results_file}. Score: {q['quality_score']:.1f}")
        return 0
    q = QualityEvaluator(a.endpoint, a.model, a.tasks, a.samples, a.api_key).evaluate()
    out = a.output or "quality_results.json"
    write_json(out, q)
    print(f"Quality evaluation complete. Score: {q['quality_score']:.1f}\nResults saved to {out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
