"""Text -> label adapter ("decoder-to-label" mapping, reference README.md:28-29; there is NO code for it in the reference).

A generated answer string is normalised and looked up in a growing answer vocabulary capped at `num_classes`; answers that
arrive after the vocabulary is full map to the last class ("other").  The normaliser is build-side and deliberately small:
lower-case, tabs/newlines -> space, punctuation removed (or turned into a space between alphanumerics), periods that are not
decimal points dropped, English articles dropped, number words zero..ten -> digits.  It follows the *order* of operations of
the VQA accuracy normaliser the reference ships for scoring (common/vqa_tools/vqa_eval.py:211-216 answer clean-up, :249-259
punctuation, :261-274 digits/articles) but does NOT carry its contraction table; tests/golden/label_adapter_golden.json holds
outputs of the reference's own normaliser on answers without contractions, which this one must reproduce.
"""
import re

_ARTICLES = {"a", "an", "the"}
_NUMBERS = {"none": "0", "zero": "0", "one": "1", "two": "2", "three": "3", "four": "4", "five": "5", "six": "6",
            "seven": "7", "eight": "8", "nine": "9", "ten": "10"}
_PUNCT = ";/[]\"{}()=+\\_-><@`,?!"
_PERIOD = re.compile(r"(?!<=\d)(\.)(?!\d)")
_COMMA_NUM = re.compile(r"(\d)(,)(\d)")


def normalize_answer(text: str) -> str:
    t = text.replace("\n", " ").replace("\t", " ").strip()
    out = t
    for p in _PUNCT:
        if (p + " " in t or " " + p in t) or _COMMA_NUM.search(t) is not None:
            out = out.replace(p, "")
        else:
            out = out.replace(p, " ")
    out = _PERIOD.sub("", out)
    words = [_NUMBERS.get(w, w) for w in out.lower().split()]
    return " ".join(w for w in words if w not in _ARTICLES)


class AnswerLabelMap:
    """Growing {normalised answer: class id}; class ids are dense in order of first appearance; id num_classes-1 is 'other'
    once the vocabulary is full.  The same map must be used for every noisy sample of an image (and across images of a run),
    so that votes are comparable."""

    def __init__(self, num_classes: int, vocabulary=()):
        assert num_classes >= 2
        self.num_classes = num_classes
        self.to_id = {}
        self.answers = []
        for a in vocabulary:
            self(a)

    def __call__(self, text: str) -> int:
        key = normalize_answer(text)
        idx = self.to_id.get(key)
        if idx is None:
            if len(self.answers) < self.num_classes - 1:
                idx = len(self.answers)
                self.to_id[key] = idx
                self.answers.append(key)
            else:
                idx = self.num_classes - 1
        return idx

    def one_hot_logits(self, texts, device=None):
        """[B, num_classes] float32 logits (1 at the label) so that a text-generating VLM can serve as Smooth's
        base_classifier: `lambda batch: label_map.one_hot_logits(model.generate(batch, ...), batch.device)`."""
        import torch
        ids = torch.tensor([self(t) for t in texts], dtype=torch.long, device=device)
        return torch.nn.functional.one_hot(ids, self.num_classes).float()
