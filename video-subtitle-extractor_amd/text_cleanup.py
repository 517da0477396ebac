"""SRT text clean-up after extraction — the rest of row N4 (backend/tools/reformat.py:16-214, switched on by
config.wordSegmentation, backend/main.py:181-182): typo replacements, re-spacing of run-together English words along a word
segmentation, and a chain of punctuation / spacing rewrites, block by block over the finished SRT.

The reference takes the word segmentation from the third-party `wordsegment` package (its unigram / bigram corpus is not
installed on either box).  Here the segmenter is a PARAMETER: `segment(text) -> [lower-case words]`; with none given,
`wordsegment.Segmenter` is imported and a missing package is an error, not a silent skip.  Everything around it is pinned by
tests/golden/text_cleanup.json — the reference's own reformat.execute run on scripted SRT files with a scripted segmenter.
"""
import json
import os
import re

# contractions the segmenter returns without their apostrophe (reformat.py:38-45): "dont" may also match "don't" in the text
_CONTRACTIONS = ["I'm", "you're", "he's", "she's", "we're", "it's", "isn't", "aren't", "they're", "there's", "wasn't", "weren't",
                 "I've", "you've", "we've", "they've", "hasn't", "haven't", "I'd", "you'd", "he'd", "she'd", "it'd", "we'd", "they'd",
                 "doesn't", "don't", "didn't", "I'll", "you'll", "he'll", "she'll", "we'll", "they'll", "there'll", "there'd",
                 "can't", "couldn't", "daren't", "hadn't", "mightn't", "mustn't", "needn't", "oughtn't", "shan't", "shouldn't",
                 "usedn't", "won't", "wouldn't", "that's", "what's", "it'll"]
_BARE = {c.replace("'", "").lower(): c for c in _CONTRACTIONS}

# the reference's backend/configs/typoMap.json at v2.2.0 (data): case-insensitive regex -> replacement, applied in this order
DEFAULT_TYPO_MAP = {"l'm": "I'm", "l just": "I just", "Let'sqo": "Let's go", "Iife": "life", "威筋": "威胁"}

# final rewrites (reformat.py:163-192), in order: (pattern, replacement) through re.sub, or (old, new, None) through str.replace
_TAIL = [
    ("([^\\sA-Z\\-])([A-Z])", "\\1 \\2"),            # a space in front of a capital that follows a non-capital
    ("  ", " ", None),
    ("。", ".", None),
    (" *([\\.\\?\\!\\,])", "\\1"),                     # no spaces in front of . ? ! ,
    (" *([\\']) *", "\\1"),                           # none around an apostrophe
    ("\n\\s*", "\n"),                                 # none at the start of a continuation line
    ("^\\s*", ""),
    ("([A-Za-z0-9]) (\\-[A-Za-z0-9])", "\\1\\2"),
    ("([A-Za-z0-9]) %", "\\1%"),
    ("·$", "."),
    (r"\bDr\. *\b", "Dr."),
    (r'[""]', '"'),
    (r"，", ","),
    ("([\\.,\\!\\?])([A-Za-z0-9\\u4e00-\\u9fa5])", "\\1 \\2"),      # a space behind . , ! ? when text follows
    ("\n\n", "\n", None),
]


def _typo_fix(text, typo_map):
    for k, v in typo_map.items():
        try:
            text = re.sub(re.compile(k, re.I), v, text)
        except re.error:
            pass
    return text


def _alts(seg_list):
    """segments -> [[word] | [word, contraction]]"""
    return [[s, _BARE[s]] if s in _BARE else [s] for s in seg_list]


def _pat(s):
    return f"({s[0]}|{s[1]})" if len(s) > 1 else f"({s[0]})"


def _keep_locatable(seg, text):
    """reformat.py:76-107: walk the segments backwards, cutting the text at the LAST match of each; keep a segment when its
    match lies in front of the previous kept one (segments the text does not contain in order are dropped)."""
    span = None
    kept = []
    for s in reversed(seg):
        hits = list(re.finditer(re.compile(_pat(s), re.I), text))
        if not hits:
            continue
        m = hits[-1]
        text = text[:m.span()[0]]
        if span is None or span > m.span():
            kept.append(s)
            span = m.span()
    return list(reversed(kept))


def cleanup_text(text, lang, segment, typo_map=None):
    """One SRT block's text -> cleaned text (reformat.py:113-196).  Returns the input unchanged where the reference skips
    (empty text; more than 1000 characters after the typo pass keeps the typo-fixed text, like the reference)."""
    typo_map = DEFAULT_TYPO_MAP if typo_map is None else typo_map
    if not text:
        return text
    text = _typo_fix(text, typo_map)
    if len(text) > 1000:
        return text
    seg = segment(text)
    if len(seg) == 1:
        seg = segment(re.sub(re.compile("(\ni)([^\\s])", re.I), "\\1 \\2", text))
    seg = _alts(seg)
    text = re.sub(" +([\\u4e00-\\u9fa5])", " \\1", text)
    if lang in ("ch", "ch_tra"):
        text = text.replace("  ", "\n")                 # two spaces separate the Chinese and the English line
    seg = _keep_locatable(seg, text)
    pieces, remain = [], text
    for i, s in enumerate(seg):
        last = i == len(seg) - 1
        m = re.search(re.compile("(.*?)" + _pat(s), re.I), remain)
        if m is None:
            if last:
                pieces.append(remain.strip())
            continue
        pieces.append(remain[:m.span()[1]].strip())
        remain = remain[m.span()[1]:].strip()
        if last:
            pieces.append(remain)
    out = " ".join(pieces) if seg else remain
    out = _typo_fix(out, typo_map)
    for rule in _TAIL:
        out = out.replace(rule[0], rule[1]) if len(rule) == 3 else re.sub(rule[0], rule[1], out)
    return out.strip()


# ---- SRT container (what the reference delegates to pysrt) ------------------------------------------------------------
_TIME = re.compile(r"^\d\d:\d\d:\d\d,\d{3} --> \d\d:\d\d:\d\d,\d{3}")


def parse_srt(data):
    """-> [(index line, time line, text)]; text = the block's remaining lines joined by \\n."""
    blocks = []
    lines = data.replace("\r\n", "\n").split("\n")
    i = 0
    while i < len(lines):
        if i + 1 < len(lines) and lines[i].strip().isdigit() and _TIME.match(lines[i + 1]):
            j = i + 2
            body = []
            while j < len(lines) and lines[j] != "":
                body.append(lines[j])
                j += 1
            blocks.append((lines[i].strip(), lines[i + 1], "\n".join(body)))
            i = j
        else:
            i += 1
    return blocks


def format_srt(blocks):
    return "".join(f"{idx}\n{tm}\n{text}\n\n" for idx, tm, text in blocks)


def default_segmenter():
    try:
        import wordsegment
    except ImportError as e:
        raise RuntimeError("text clean-up needs a word segmenter: install `wordsegment` (the reference's dependency) or pass "
                           "segment=callable(text) -> list of lower-case words") from e
    seg = wordsegment.Segmenter()
    seg.load()
    return seg.segment


def cleanup_srt(srt_text, lang, segment, typo_map=None):
    """The same over SRT text in memory -> (new SRT text, number of modified blocks)."""
    out, modified = [], 0
    for idx, tm, text in parse_srt(srt_text):
        try:
            new = cleanup_text(text, lang, segment, typo_map)
        except Exception:               # reformat.py keeps a block it cannot process (segmenter / regex failure) and goes on
            new = text
        modified += int(new != text)
        out.append((idx, tm, new))
    return format_srt(out), modified


def execute(path, lang="en", segment=None, typo_map=None):
    """reformat.execute(path, lang): rewrites the SRT file in place; -> True like the reference (the number of modified
    blocks is available from cleanup_srt)."""
    if segment is None:
        segment = default_segmenter()
    if typo_map is None:
        p = os.environ.get("VSE_TYPO_MAP")
        typo_map = json.load(open(p, encoding="utf-8")) if p else DEFAULT_TYPO_MAP
    with open(path, encoding="utf-8") as f:
        new, modified = cleanup_srt(f.read(), lang, segment, typo_map)
    with open(path, "w", encoding="utf-8") as f:
        f.write(new)
    return True
