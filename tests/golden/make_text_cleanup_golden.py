#!/usr/bin/env python3
"""Generates tests/golden/text_cleanup.json by RUNNING the reference's backend/tools/reformat.py (execute) in this container
on scripted SRT files.  Its two third-party imports are replaced by stubs that carry behaviour, and are therefore part of
what the vectors pin:
  * wordsegment.Segmenter -> a scripted segmenter (clean to [a-z0-9], greedy longest match over a small vocabulary) — the real
    package's corpus is not installed; the product takes the segmenter as a parameter and is run with the same one;
  * pysrt.open / save -> a minimal SRT reader / writer (index, time line, text joined by newlines; "idx\\ntimes\\ntext\\n\\n").
The typo map is the reference's own backend/configs/typoMap.json.  Only inputs and outputs are written (data, not source).
Needs /root/reference; not run on the GPU box."""
import importlib.util
import json
import os
import re
import sys
import tempfile
import types

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "text_cleanup.json")
VOCAB = sorted(["i", "im", "a", "am", "the", "this", "that", "thats", "is", "isnt", "it", "its", "was", "what", "whats", "you",
                "youre", "we", "well", "they", "dont", "cant", "wont", "do", "not", "go", "going", "lets", "let", "s", "just",
                "life", "hello", "world", "good", "morning", "doctor", "dr", "smith", "said", "ten", "percent", "of", "people",
                "like", "self", "made", "men", "and", "women", "to", "be", "or", "here", "there", "now", "how", "are", "my",
                "name", "john", "new", "york", "city", "love", "one", "two", "10", "20", "2021"], key=len, reverse=True)


def scripted_segment(text):
    s = re.sub("[^a-z0-9]", "", text.lower())
    out, i = [], 0
    while i < len(s):
        for w in VOCAB:
            if s.startswith(w, i):
                out.append(w)
                i += len(w)
                break
        else:
            out.append(s[i])
            i += 1
    return out


class _Item:
    def __init__(self, idx, tm, text):
        self.index, self.tm, self.text = idx, tm, text


class _Subs(list):
    def save(self, path, encoding="utf-8"):
        with open(path, "w", encoding=encoding) as f:
            for it in self:
                f.write(f"{it.index}\n{it.tm}\n{it.text}\n\n")


def _open(path, encoding="utf-8"):
    data = open(path, encoding=encoding).read().replace("\r\n", "\n")
    subs = _Subs()
    for blk in data.split("\n\n"):
        lines = blk.split("\n")
        while lines and lines[0] == "":
            lines.pop(0)
        if len(lines) >= 2 and lines[0].strip().isdigit() and "-->" in lines[1]:
            subs.append(_Item(lines[0].strip(), lines[1], "\n".join(lines[2:])))
    return subs


def install():
    class Segmenter:
        def load(self):
            pass

        def segment(self, text):
            return scripted_segment(text)
    ws = types.ModuleType("wordsegment")
    ws.Segmenter = Segmenter
    sys.modules["wordsegment"] = ws
    ps = types.ModuleType("pysrt")
    ps.open = _open
    sys.modules["pysrt"] = ps
    spec = importlib.util.spec_from_file_location("ref_reformat", os.path.join(REF, "backend", "tools", "reformat.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


TEXTS = [
    "helloworld", "Hello world", "l'm going", "Iife is good", "Let'sqo now", "thisisit", "dontgo", "I dont like it",
    "whatsyourname", "Hello,world!How are you?", "He said:\"hello\" ，ok", "tenpercent of people", "10 % of people",
    "self -made men", "Dr. Smith said hello", "goodMorningNewYork", "你好世界  helloworld", "你好   世界", "威筋 is here",
    "hello 。world", "thats what i said ·", "it 's mine", "line one\nlinetwo", "line one\n  i am here", "x", "",
    "well  they wont go ,  will they ?", "NEWYORKCITY", "i love new york-city", "2021 was good", "a" * 1001,
    "mynameisjohn\nandthisisit", "this is it.this is not", "HELLO world", "hello\n\nworld",
]


def make_srt(texts):
    out = []
    for i, t in enumerate(texts):
        out.append(f"{i + 1}\n00:00:{i:02d},000 --> 00:00:{i:02d},900\n{t}\n\n")
    return "".join(out)


def main():
    m = install()
    typo = json.load(open(os.path.join(REF, "backend", "configs", "typoMap.json"), encoding="utf-8"))
    cases = []
    for lang in ("en", "ch"):
        for chunk in (TEXTS[:12], TEXTS[12:24], TEXTS[24:]):
            src = make_srt([t for t in chunk])
            with tempfile.TemporaryDirectory() as td:
                p = os.path.join(td, "x.srt")
                open(p, "w", encoding="utf-8").write(src)
                ok = m.execute(p, lang)
                cases.append({"lang": lang, "input": src, "ok": bool(ok), "output": open(p, encoding="utf-8").read()})
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump({"source": "backend/tools/reformat.py:16-214 @ v2.2.0 executed with a scripted wordsegment and a minimal pysrt",
                   "vocab": VOCAB, "typo_map": typo, "cases": cases}, f, ensure_ascii=False, separators=(",", ":"))
    print("wrote", OUT, len(cases), "files,", sum(c["input"].count("-->") for c in cases), "blocks")


if __name__ == "__main__":
    main()
