"""raw.txt clean-up between OCR and SRT writing (the part that needs no absent corpus): the
station-logo ("watermark") filter and the scene-text filter the reference runs when no subtitle area was given.

Follows backend/main.py of video-subtitle-extractor v2.2.0:
  * unite_coordinates      <- SubtitleExtractor._unite_coordinates + __is_coordinate_similar   (main.py:866-881,954-963)
  * detect_watermark_area  <- SubtitleExtractor._detect_watermark_area                        (main.py:671-711)
  * detect_subtitle_area   <- SubtitleExtractor._detect_subtitle_area                         (main.py:713-729)
  * filter_watermark       <- SubtitleExtractor.filter_watermark                              (main.py:506-565)
  * filter_scene_text      <- SubtitleExtractor.filter_scene_text                             (main.py:567-612)
The reference asks the user on stdin (and draws the candidate areas on a random frame for them to look at); here the
decision is a callback `decide(area) -> bool`, everything else is the same file transformation.  Pinned by
tests/golden/raw_filters.json (tests/golden/make_raw_filters_golden.py ran the reference's methods with scripted
answers).  Quirks kept: coordinates are united in place while the list is being walked (later rows see earlier
replacements, the LAST similar row wins), a text containing a tab loses everything after it (and its newline) when the
watermark detector rewrites the file, a watermark line is dropped when the area's repr occurs ANYWHERE in the line, the
scene-text band is abs(ymin - dev) .. ymax + dev, ties in the frequency counts keep first-seen order.
tolerantPixelX / tolerantPixelY are compared as plain numbers (backend/config.py:66-67: 100 / 50).
"""
import io
from collections import Counter

import numpy as np

TOLERANT_PIXEL_X = 100      # backend/config.py:67
TOLERANT_PIXEL_Y = 50       # backend/config.py:66
SUBTITLE_AREA_DEVIATION_PIXEL = 50   # backend/config.py:69
WATERMARK_AREA_NUM = 5      # backend/config.py:71


def unite_coordinates(coords, tol_x=TOLERANT_PIXEL_X, tol_y=TOLERANT_PIXEL_Y):
    """list of (xmin, xmax, ymin, ymax) -> same list object with similar boxes replaced by one representative.
    Row `k` becomes the last row (in the list's state at that moment) similar to the ORIGINAL row k; one numpy pass
    per row instead of the reference's Python double loop, same result."""
    n = len(coords)
    if n == 0:
        return coords
    cur = np.asarray(coords, dtype=np.int64).reshape(n, 4)
    tol = np.array([tol_x, tol_x, tol_y, tol_y], dtype=np.int64)
    for k in range(n):
        me = cur[k].copy()
        js = np.nonzero((np.abs(cur - me) < tol).all(axis=1))[0]
        if js.size == 0:
            continue                                    # tolerance 0: not even similar to itself
        last = js[-1]
        if last > k:
            cur[k] = cur[last]
        else:
            # the walk reaches row k itself last: by then it holds the last similar row before it (if any), and a
            # row that is not similar to itself (cannot happen for tol >= 1) keeps that replacement too
            prev = js[js < k]
            if prev.size:
                cur[k] = cur[prev[-1]]
    for k in range(n):
        coords[k] = tuple(int(v) for v in cur[k])
    return coords


def _coords(line):
    pos = line.split('\t')[1].split('(')[1].split(')')[0].split(', ')
    return int(pos[0]), int(pos[1]), int(pos[2]), int(pos[3])


def _parse(line):
    parts = line.split('\t')
    return parts[0], _coords(line), parts[2]


def detect_watermark_area(lines, num=WATERMARK_AREA_NUM, tol_x=TOLERANT_PIXEL_X, tol_y=TOLERANT_PIXEL_Y):
    """raw lines -> ([(area, count), ...] most frequent first, at most `num`; the rewritten raw lines)."""
    rows = [_parse(ln) for ln in lines]
    coords = unite_coordinates([r[1] for r in rows], tol_x, tol_y)
    # the reference rewrites raw.txt here and every later step reads the FILE back: a text that contained a tab has lost its
    # tail and its newline, so it now shares one line with the row that follows it (found by fuzzing against the reference)
    out = io.StringIO("".join(f'{r[0]}\t{c}\t{r[2]}' for r, c in zip(rows, coords))).readlines()
    common = Counter(coords).most_common()
    return (common[:num] if len(common) > num else common), out


def detect_subtitle_area(lines):
    """raw lines -> [((ymin, ymax), count)] of the most frequent vertical extent ([] for an empty file)."""
    return Counter(_coords(ln)[2:] for ln in lines).most_common(1)


def filter_watermark(lines, decide, num=WATERMARK_AREA_NUM, tol_x=TOLERANT_PIXEL_X, tol_y=TOLERANT_PIXEL_Y):
    """Drops the lines of every candidate area for which decide((area, count)) is true."""
    areas, lines = detect_watermark_area(lines, num, tol_x, tol_y)
    for area in areas:
        if decide(area):
            key = str(area[0])
            lines = [ln for ln in lines if ln.find(key) == -1]
    return lines


def filter_scene_text(lines, decide=lambda band: True, deviation=SUBTITLE_AREA_DEVIATION_PIXEL):
    """Keeps only the lines whose vertical extent lies inside the most frequent one widened by `deviation`."""
    area = detect_subtitle_area(lines)[0][0]             # IndexError on an empty file, like the reference
    ymin = abs(area[0] - deviation)
    ymax = area[1] + deviation
    if not decide((ymin, ymax)):
        return lines
    keep = []
    for ln in lines:
        c = _coords(ln)
        if ymin <= c[2] and c[3] <= ymax:
            keep.append(ln)
    return keep


def _rewrite(path, fn):
    with open(path, mode='r', encoding='utf-8') as f:
        lines = f.readlines()
    lines = fn(lines)
    with open(path, mode='w', encoding='utf-8') as f:
        f.writelines(lines)


def filter_watermark_file(raw_path, decide, **kw):
    _rewrite(raw_path, lambda lines: filter_watermark(lines, decide, **kw))


def filter_scene_text_file(raw_path, decide=lambda band: True, **kw):
    _rewrite(raw_path, lambda lines: filter_scene_text(lines, decide, **kw))
