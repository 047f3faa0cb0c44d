r"""Text -> label adapter ("decoder-to-label" mapping, reference README.md:28-29; there is NO code for it in the reference).

A generated answer string is normalised and looked up in an answer vocabulary of at most `num_classes - 1` entries; the last
class id is "other".  The normaliser reproduces the VQA accuracy normaliser the reference ships for scoring
(common/vqa_tools/vqa_eval.py:211-216 answer clean-up, :249-259 punctuation, :261-274 number words / articles / contractions),
including its quirks (the `(?!<=\d)` look-ahead typo, the period strip that stops after 32 matches, the reversed "somebody'd" entry, and the fact that its capitalised
contraction keys can never match a lower-cased word).  tests/golden/label_adapter_golden.json holds outputs of the reference's
own normaliser -- every key of its contraction table among them -- which this one must reproduce.

The contraction table is generated from its grammar (stem x suffix) instead of being listed entry by entry.
"""
import re

_ARTICLES = {"a", "an", "the"}
_NUMBERS = {"none": "0", "zero": "0", "one": "1", "two": "2", "three": "3", "four": "4", "five": "5", "six": "6",
            "seven": "7", "eight": "8", "nine": "9", "ten": "10"}
_PUNCT = ";/[]\"{}()=+\\_-><@`,?!"
_PERIOD = re.compile(r"(?!<=\d)(\.)(?!\d)")
_COMMA_NUM = re.compile(r"(\d)(,)(\d)")


def _build_contractions():
    """apostrophe-less (or half-apostrophised) word -> contraction, as vqa_eval.py:29-150 defines it."""
    t = {}
    for stem in ("ai", "are", "ca", "could", "did", "does", "do", "had", "has", "have", "is", "might", "must", "need", "ought",
                 "sha", "should", "was", "were", "wo", "would"):
        t[stem + "nt"] = stem + "n't"
    for stem in ("could", "had", "might", "should", "would"):
        t[stem + "nt've"] = t[stem + "n'tve"] = stem + "n't've"
    for stem in ("could", "might", "must", "should", "would", "not", "they", "we", "who", "you", "what", "where"):
        t[stem + "ve"] = stem + "'ve"
    for stem in ("he", "how", "it", "they", "where", "who", "you", "someone", "something", "there"):
        t[stem + "d"] = stem + "'d"
    for stem in ("he", "it", "she", "somebody", "someone", "something", "there", "they", "we", "who", "you"):
        t[stem + "d've"] = t[stem + "'dve"] = stem + "'d've"
    for stem in ("how", "it", "somebody", "someone", "something", "they", "what", "who", "why", "you"):
        t[stem + "ll"] = stem + "'ll"
    for stem in ("he", "how", "somebody", "someone", "that", "there", "what", "when", "where", "who", "why"):
        t[stem + "s"] = stem + "'s"
    for stem in ("there", "they", "what", "why", "you"):
        t[stem + "re"] = stem + "'re"
    t.update({"maam": "ma'am", "oclock": "o'clock", "twas": "'twas", "yall": "y'all", "let's": "let's", "she's": "she's",
              "yall'll": "y'all'll", "y'allll": "y'all'll", "yall'd've": "y'all'd've", "y'alld've": "y'all'd've",
              "y'all'dve": "y'all'd've", "ow's'at": "'ow's'at", "'ows'at": "'ow's'at", "'ow'sat": "'ow's'at",
              "somebody'd": "somebodyd"})        # the last one is the reference's own (reversed) entry
    return t


_CONTRACTIONS = _build_contractions()


def _process_punctuation(t: str) -> str:
    """vqa_eval.py:249-259."""
    out = t
    for p in _PUNCT:
        if (p + " " in t or " " + p in t) or _COMMA_NUM.search(t) is not None:
            out = out.replace(p, "")
        else:
            out = out.replace(p, " ")
    return _PERIOD.sub("", out, count=32)        # the reference passes re.UNICODE (= 32) in the COUNT position, vqa_eval.py:257


def normalize_answer(text: str) -> str:
    """The clean-up the reference applies to a PREDICTED answer before scoring (vqa_eval.py:211-216 + :249-274)."""
    out = _process_punctuation(text.replace("\n", " ").replace("\t", " ").strip())
    words = [_NUMBERS.get(w, w) for w in out.lower().split()]
    return " ".join(_CONTRACTIONS.get(w, w) for w in words if w not in _ARTICLES)


def vqa_accuracy(answer: str, gt_answers) -> float:
    """VQA accuracy of one predicted answer against the (ten) human answers of a question, as the reference's evaluation loop
    computes it (vqa_eval.py:211-247, the scoring behind agents/minigpt4_eval_agent.py:108-113): the prediction is normalised;
    the ground-truth answers only have their punctuation processed, and only when they are not all identical; for each ground-truth
    answer the prediction scores min(1, matches among the OTHER answers / 3), averaged over the answers.  Pinned by goldens emitted
    by the reference's own VQAEval.evaluate (tests/golden/label_adapter_golden.json, `vqa_accuracy`)."""
    res = normalize_answer(answer)
    gts = list(gt_answers)
    if not gts:
        return 0.0
    if len(set(gts)) > 1:
        gts = [_process_punctuation(g) for g in gts]
    total = 0.0
    for i in range(len(gts)):
        matches = sum(1 for j, g in enumerate(gts) if j != i and g == res)
        total += min(1.0, matches / 3.0)
    return total / len(gts)


class AnswerLabelMap:
    """{normalised answer: class id}.  Class ids 0 .. num_classes-2 are answers in vocabulary order; id num_classes-1 is
    "other" (`other_id`): every answer outside the vocabulary.  "other" is not a class of the base classifier in the sense of
    the certificate (it lumps unrelated answers together), so callers treat a top class of `other_id` as ABSTAIN
    (`Smooth(..., non_certifiable=(label_map.other_id,))`).

    The base classifier must be a FIXED function of the image for the certificate to hold, so the vocabulary should be given
    up front and the map frozen.  An unfrozen map grows in order of first appearance (single process, exploratory use only):
    under torch.distributed with more than one rank that would give the same answer different ids on different ranks (each
    rank sees different noisy samples) and the all-reduce would sum unrelated classes -- growing there raises."""

    def __init__(self, num_classes: int, vocabulary=(), frozen=None):
        assert num_classes >= 2
        self.num_classes = num_classes
        self.to_id = {}
        self.answers = []
        self.frozen = False
        for a in vocabulary:                                   # given up front: the same on every rank by construction
            key = normalize_answer(a)
            if key not in self.to_id and len(self.answers) < self.num_classes - 1:
                self.to_id[key] = len(self.answers)
                self.answers.append(key)
        # a vocabulary given up front freezes the map unless the caller says otherwise
        self.frozen = bool(vocabulary) if frozen is None else bool(frozen)

    @property
    def other_id(self) -> int:
        return self.num_classes - 1

    def freeze(self):
        self.frozen = True
        return self

    @staticmethod
    def _multi_rank() -> bool:
        try:
            import torch.distributed as dist
            return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        except Exception:
            return False

    def __call__(self, text: str) -> int:
        key = normalize_answer(text)
        idx = self.to_id.get(key)
        if idx is None:
            if self.frozen or len(self.answers) >= self.num_classes - 1:
                return self.other_id
            if self._multi_rank():
                raise RuntimeError("AnswerLabelMap would grow under torch.distributed (ids would differ between ranks): "
                                   "pass the answer vocabulary up front (it is then frozen)")
            idx = len(self.answers)
            self.to_id[key] = idx
            self.answers.append(key)
        return idx

    def one_hot_logits(self, texts, device=None):
        """[B, num_classes] float32 logits (1 at the label) so that a text-generating VLM can serve as Smooth's
        base_classifier: `lambda batch: label_map.one_hot_logits(model.generate(batch, ...), batch.device)`."""
        import torch
        ids = torch.tensor([self(t) for t in texts], dtype=torch.long, device=device)
        return torch.nn.functional.one_hot(ids, self.num_classes).float()
