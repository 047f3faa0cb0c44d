#!/usr/bin/env python3
"""Golden vectors for the text->label normaliser: outputs of the REFERENCE's own VQA answer normaliser
(common/vqa_tools/vqa_eval.py: the clean-up at :211-216, processPunctuation :249-259, processDigitArticle :261-274) on a list
of answer strings (`cases`), among them every key of its contraction table (as a lone word and inside a sentence) and the contracted
forms themselves; and (`vqa_accuracy`) the VQA accuracy of a predicted answer against ten ground-truth answers as the
reference's own VQAEval.evaluate computes it (vqa_eval.py:196-247).  Build container only; only the JSON travels.

    python oracle/gen_golden_labels.py
"""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ref_vqa_eval", "/root/reference/common/vqa_tools/vqa_eval.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
ev = mod.VQAEval()          # vqa=None: only the tables / regexes are built (vqa_eval.py:19-191)

ANSWERS = ["Yes", "no.", " A dog ", "the red car", "Two", "two dogs and a cat", "1,000", "3.5", "It is a table.", "blue; green",
           "on the left/right", "a man (standing)", "What?", "yes!", "an apple, a pear", "THE END", "ten people", "none",
           "zero", "skate-board", "tennis racket", "10", "he is 5 feet tall", "black & white", "pizza\nand fries", "a\tb",
           "one two three", "stop sign.", "U.S.A.", "2.", ".5", "green,blue", "frisbee?!", "left", "right side of the road"]

ANSWERS += ["I dont know", "its not there, isnt it", "thats what shes doing", "Im sure Ive seen it", "somebody'd say so",
            "theyre at 5 oclock", "yall come back", "he said: couldnt've", "whats that?", "don't", "it's"]
# periodStrip.sub("", outText, re.UNICODE) (vqa_eval.py:257): the third positional argument is COUNT (= int(re.UNICODE) = 32), so only
# the first 32 matching periods are stripped
ANSWERS += ["a" + "." * 31 + "b", "a" + "." * 32 + "b", "a" + "." * 33 + "b", "x. " * 40 + "end.", "..." * 15 + " 3.5 " + ". ." * 10]
ANSWERS += sorted(ev.contractions.keys())
ANSWERS += ["the " + k + " thing" for k in sorted(ev.contractions.keys())[::7]]
ANSWERS += sorted(set(ev.contractions.values()))

out = []
for a in ANSWERS:
    r = a.replace("\n", " ").replace("\t", " ").strip()
    r = ev.processPunctuation(r)
    r = ev.processDigitArticle(r)
    out.append({"answer": a, "normalized": r})
# ---- VQA accuracy of one predicted answer against ten ground-truth answers: the reference's own VQAEval.evaluate (vqa_eval.py:196-247)
# run on stand-in VQA objects that expose exactly what it touches (getQuesIds, .qa); one question per case.
class _VQA:
    def __init__(self, qa):
        self.qa = qa

    def getQuesIds(self):
        return list(self.qa.keys())


def ref_accuracy(answer, gts):
    import io, contextlib
    gt = {1: {"answers": [{"answer": a, "answer_confidence": "yes", "answer_id": i + 1} for i, a in enumerate(gts)],
              "question_type": "what", "answer_type": "other", "image_id": 1, "question_id": 1}}
    res = {1: {"answer": answer, "question_id": 1}}
    e = mod.VQAEval(_VQA(gt), _VQA(res), n=6)
    with contextlib.redirect_stdout(io.StringIO()):
        e.evaluate()
    return e.evalQA[1] / 100.0


ACC_CASES = [
    ("yes", ["yes"] * 10), ("Yes.", ["yes"] * 10), ("no", ["yes"] * 10), ("yes", ["yes"] * 3 + ["no"] * 7), ("yes", ["yes"] * 2 + ["no"] * 8),
    ("yes", ["yes"] * 1 + ["no"] * 9), ("no", ["yes"] * 3 + ["no"] * 7), ("two", ["2"] * 6 + ["two"] * 4), ("2", ["2"] * 6 + ["two"] * 4),
    ("a dog", ["dog"] * 5 + ["a dog"] * 5), ("dog", ["dog"] * 5 + ["a dog"] * 5), ("the dog", ["dog"] * 10),
    ("dont know", ["don't know"] * 4 + ["unknown"] * 6), ("don't know", ["don't know"] * 4 + ["unknown"] * 6),
    ("red, white", ["red white"] * 4 + ["red, white"] * 3 + ["white"] * 3), ("skate-board", ["skateboard"] * 5 + ["skate board"] * 5),
    ("frisbee?", ["frisbee"] * 9 + ["disc"]), ("3.5", ["3.5"] * 2 + ["4"] * 8), ("1,000", ["1000"] * 4 + ["1,000"] * 6),
    ("blue", ["blue", "blue", "blue", "navy", "navy", "dark blue", "dark blue", "teal", "aqua", "blue"]),
    ("navy", ["blue", "blue", "blue", "navy", "navy", "dark blue", "dark blue", "teal", "aqua", "blue"]),
    ("", ["yes"] * 10), ("ten people", ["10 people"] * 3 + ["ten people"] * 3 + ["many"] * 4), ("U.S.A.", ["usa"] * 5 + ["u.s.a."] * 5),
    ("w17 w5", ["w17 w5"] * 4 + ["w3"] * 6), ("maam", ["ma'am"] * 10),
]
acc = [{"answer": a, "gt_answers": g, "accuracy": ref_accuracy(a, g)} for a, g in ACC_CASES]

path = os.path.join(ROOT, "tests", "golden", "label_adapter_golden.json")
json.dump({"generator": "oracle/gen_golden_labels.py", "cases": out, "vqa_accuracy": acc}, open(path, "w"), indent=0)
print("wrote", path, len(out), len(acc))
