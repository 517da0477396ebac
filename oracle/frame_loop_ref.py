"""ORACLE (test infrastructure, not product code) — CPU restatement of the reference's accurate-mode frame loop.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows backend/main.py:255-376 (extract_frame_by_det), :906-922 (__get_area_text), :924-952 (_compare_ocr_result)
statement by statement, with the detector, recogniser and frame source injected.  PINNED by
tests/golden/frame_loop.json, produced by executing the reference's own code on scripted inputs
(tests/golden/make_frame_loop_golden.py).  `ratio` restates Levenshtein==0.26.0's ratio (requirements.txt:2):
normalised InDel similarity 2*LCS/(len(a)+len(b)), 1.0 for two empty strings.
"""


def ratio(a, b):
    if not a and not b:
        return 1.0
    prev = [0] * (len(b) + 1)
    for ca in a:
        cur = [0]
        for j, cb in enumerate(b):
            cur.append(prev[j] + 1 if ca == cb else max(prev[j + 1], cur[j]))
        prev = cur
    return 2.0 * prev[-1] / (len(a) + len(b))


def get_coordinates(dt_box):
    """backend/tools/ocr.py:115-134."""
    out = []
    if isinstance(dt_box, list):
        for i in dt_box:
            i = list(i)
            x1, y1 = int(i[0][0]), int(i[0][1])
            x2, y2 = int(i[1][0]), int(i[1][1])
            x3, y3 = int(i[2][0]), int(i[2][1])
            x4, y4 = int(i[3][0]), int(i[3][1])
            out.append((max(x1, x4), min(x2, x3), max(y1, y2), min(y3, y4)))
    return out


def _area_text(ocr_result, sub_area):
    """backend/main.py:906-922."""
    box, text = ocr_result
    out = []
    for content, c in zip(text, get_coordinates(box)):
        if sub_area is not None:
            if sub_area["xmin"] <= c[0] and c[1] <= sub_area["xmax"] and sub_area["ymin"] <= c[2] and c[3] <= sub_area["ymax"]:
                out.append(content[0])
    return out


def _compare(cache, predict, sub_area, img1, img1_no, img2, img2_no, threshold):
    """backend/main.py:924-952."""
    for img, no in ((img1, img1_no), (img2, img2_no)):
        if no not in cache:
            dt_box, rec_res = predict(img)
            cache[no] = {"text": "".join(_area_text((dt_box, rec_res), sub_area)), "dt_box": dt_box, "rec_res": rec_res}
    t1, t2 = cache[img1_no]["text"], cache[img2_no]["text"]
    for no in [n for n in cache if n < min(img1_no, img2_no) - 10]:
        del cache[no]
    return ratio(t1, t2) > threshold / 100.0


def extract_frame_by_det(frames, frame_count, detect, predict, sub_area, threshold=80):
    """frames: iterable of frames (1-based numbering); detect(frame) -> ndarray[N,4,2]; predict(frame) -> (boxes, res).
    Returns the task list [(frame_count, frame_no, dt_box, rec_res)] in queue order (backend/main.py:255-376)."""
    tasks = []
    current = 0
    ocr_args = []
    cache = {}
    first_flag = True
    finding_start = False
    finding_end = False
    start_no = 0
    start_frame = None

    def flush(keep):
        while len(ocr_args) > keep:
            total, no = ocr_args.pop(0)
            if current in cache:                       # sic: looked up by the CURRENT frame number (:355, :368)
                r = cache[current]
                tasks.append((total, no, r["dt_box"], r["rec_res"]))
            else:
                tasks.append((total, no, None, None))

    for frame in frames:
        current += 1
        dt_boxes = detect(frame)
        has = False
        if sub_area is not None:
            for (xmin, xmax, ymin, ymax) in get_coordinates(dt_boxes.tolist()):
                if sub_area["xmin"] <= xmin and xmax <= sub_area["xmax"] and sub_area["ymin"] <= ymin and ymax <= sub_area["ymax"]:
                    has = True
                    if first_flag:
                        finding_start = True
                        first_flag = False
                    break
        else:
            has = len(dt_boxes) > 0
        if has:
            if finding_start:
                start_no = current
                dt_box, rec_res = predict(frame)
                text1 = "".join(_area_text((dt_box, rec_res), sub_area))
                if start_no not in cache:
                    cache[current] = {"text": text1, "dt_box": dt_box, "rec_res": rec_res}
                    ocr_args.append((frame_count, current))
                    start_frame = frame
                finding_start = False
                finding_end = True
            if finding_end and current == frame_count:
                finding_end = False
                finding_start = False
                ocr_args.append((frame_count, current))
            if finding_end:
                if not _compare(cache, predict, sub_area, None, start_no, frame, current, threshold):
                    finding_end = False
                    finding_start = True
                    ocr_args.append((frame_count, current - 1))
        else:
            if finding_end:
                finding_end = False
                finding_start = True
                ocr_args.append((frame_count, current - 1))
        flush(1)
    flush(0)
    return tasks
