"""Seeded synthetic video frames for tests and the benchmark: there is no decodable
video on either box, so frames are generated: noisy dark background with a vertical gradient and 1-2 subtitle
lines (white fill, black outline) inside the reference's default subtitle area
(y in [0.78,0.99]*H, x in [0.05,0.95]*W; backend/config.py:49)."""
import numpy as np

_WORDS = ("the quick brown fox jumps over a lazy dog while seven wizards quietly box with grumpy elves "
          "near frozen lakes and bright morning light shines across silent hills").split()


def _font(size):
    from PIL import ImageFont
    for p in ("/usr/share/fonts/truetype/dejavu/DejaVuSans-Bold.ttf", "DejaVuSans-Bold.ttf"):
        try:
            return ImageFont.truetype(p, size)
        except Exception:
            continue
    return None


def render_line(text, height, rng):
    """-> (uint8 [h,w] fill mask, uint8 [h,w] outline mask)"""
    from PIL import Image, ImageDraw
    font = _font(int(height * 0.8))
    if font is None:
        # pseudo-glyph fallback: vertical strokes
        w = len(text) * height // 2
        fill = np.zeros((height, w), np.uint8)
        for i in range(0, w - 6, max(6, height // 3)):
            fill[rng.integers(2, height // 3):height - rng.integers(2, height // 3), i:i + 4] = 255
        outline = np.zeros_like(fill)
        return fill, outline
    tmp = Image.new("L", (4096, height * 2), 0)
    d = ImageDraw.Draw(tmp)
    d.text((8, height // 4), text, fill=255, font=font, stroke_width=2, stroke_fill=128)
    a = np.asarray(tmp)
    ys, xs = np.nonzero(a)
    a = a[max(ys.min() - 1, 0):ys.max() + 2, max(xs.min() - 1, 0):xs.max() + 2]
    return (a == 255).astype(np.uint8) * 255, (a == 128).astype(np.uint8) * 255


def make_frames(n, height=1080, width=1920, seed=0, p_two_lines=0.2, return_truth=False):
    """uint8 BGR [n,height,width,3] (+ list of ground-truth (x0,y0,x1,y1,text) per frame)."""
    rng = np.random.default_rng(seed)
    frames = np.empty((n, height, width, 3), np.uint8)
    truth = []
    grad = np.linspace(0, 40, height, dtype=np.float32)[:, None, None]
    scale = height / 1080.0
    for f in range(n):
        base = rng.integers(30, 91, size=((height + 3) // 4, (width + 3) // 4, 3), dtype=np.uint8)
        img = np.repeat(np.repeat(base, 4, 0), 4, 1)[:height, :width].astype(np.float32) + grad
        lines = 2 if rng.random() < p_two_lines else 1
        gh = int(rng.integers(54, 67) * scale)
        y_lo, y_hi = int(0.78 * height), int(0.99 * height)
        ys = [y_hi - gh - 8] if lines == 1 else [y_lo + 6, y_lo + gh + 26]
        tr = []
        for y in ys:
            nwords = int(rng.integers(3, 8))
            text = " ".join(rng.choice(_WORDS, nwords))
            fill, outline = render_line(text, gh, rng)
            lh, lw = fill.shape
            lw = min(lw, int(0.88 * width))
            fill, outline = fill[:, :lw], outline[:, :lw]
            x = (width - lw) // 2
            y = min(y, height - lh - 2)
            reg = img[y:y + lh, x:x + lw]
            reg[outline > 0] = 0
            reg[fill > 0] = 255
            tr.append((x, y, x + lw, y + lh, text))
        frames[f] = np.clip(img, 0, 255).astype(np.uint8)
        truth.append(tr)
    return (frames, truth) if return_truth else frames
