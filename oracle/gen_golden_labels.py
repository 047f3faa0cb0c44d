#!/usr/bin/env python3
"""Golden vectors for the text->label normaliser: outputs of the REFERENCE's own VQA answer normaliser
(common/vqa_tools/vqa_eval.py: the clean-up at :211-216, processPunctuation :249-259, processDigitArticle :261-274) on a list
of answer strings, among them every key of its contraction table (as a lone word and inside a sentence) and the contracted
forms themselves.  Build container only; only the JSON travels.

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
ANSWERS += sorted(ev.contractions.keys())
ANSWERS += ["the " + k + " thing" for k in sorted(ev.contractions.keys())[::7]]
ANSWERS += sorted(set(ev.contractions.values()))

out = []
for a in ANSWERS:
    r = a.replace("\n", " ").replace("\t", " ").strip()
    r = ev.processPunctuation(r)
    r = ev.processDigitArticle(r)
    out.append({"answer": a, "normalized": r})
path = os.path.join(ROOT, "tests", "golden", "label_adapter_golden.json")
json.dump({"generator": "oracle/gen_golden_labels.py", "cases": out}, open(path, "w"), indent=0)
print("wrote", path, len(out))
