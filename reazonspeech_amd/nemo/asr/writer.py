"""Subtitle / transcript writers (pkg/nemo-asr/src/writer.py:4-168), table-driven.

Formats and their exact byte output follow the reference (pinned by
tests/golden/reference_host.json): WebVTT, SubRip, ASS, JSON lines, TSV and the default
"[start --> end] text" lines.  `get_writer` keeps the reference's extension quirk
(writer.py:160-166): with no explicit format the file name's extension is compared WITH its
dot (".vtt") against the bare class extension ("vtt"), so it never matches and the plain text
writer is chosen — callers must pass `--to`.
"""
import json
import os


def _hms(seconds):
    return int(seconds / 3600), int(seconds / 60) % 60, int(seconds % 60)


def _clock(seconds, sep=".", frac_digits=3, hour_width=2):
    h, m, s = _hms(seconds)
    frac = int((seconds % 1) * (10 ** frac_digits))
    return "%0*i:%02i:%02i%s%0*i" % (hour_width, h, m, s, sep, frac_digits, frac)


class _Writer:
    ext = "txt"
    header = ""

    def __init__(self, fp):
        self.fp = fp

    def write_header(self):
        if self.header:
            self.fp.write(self.header)

    def write(self, segment):
        self.fp.write(self.line(segment))


class VTTWriter(_Writer):
    ext = "vtt"
    header = "WEBVTT\n\n"

    def line(self, seg):
        return "%s --> %s\n%s\n\n" % (_clock(seg.start_seconds), _clock(seg.end_seconds), seg.text)


class SRTWriter(_Writer):
    ext = "srt"

    def __init__(self, fp):
        super().__init__(fp)
        self.index = 0

    def line(self, seg):
        self.index += 1
        return "%i\n%s --> %s\n%s\n\n" % (self.index, _clock(seg.start_seconds, ","),
                                          _clock(seg.end_seconds, ","), seg.text)


class ASSWriter(_Writer):
    ext = "ass"
    header = ("[Script Info]\nScriptType: v4.00+\nCollisions: Normal\nTimer: 100.0000\n\n"
              "[V4+ Styles]\nStyle: Default,Arial,16,&Hffffff,&Hffffff,&H0,&H0,0,0,0,0,100,100,0,0,1,1,0,2,10,10,10,0\n\n"
              "[Events]\n")

    def line(self, seg):
        return "Dialogue: 0,%s,%s,Default,,0,0,0,,%s\n" % (
            _clock(seg.start_seconds, ".", 2, 1), _clock(seg.end_seconds, ".", 2, 1), seg.text)


class JSONWriter(_Writer):
    ext = "json"

    def line(self, seg):
        return json.dumps({"start_seconds": round(seg.start_seconds, 3),
                           "end_seconds": round(seg.end_seconds, 3),
                           "text": seg.text}, ensure_ascii=False) + "\n"


class TSVWriter(_Writer):
    ext = "tsv"
    header = "start_seconds\tend_seconds\ttext\n"

    def line(self, seg):
        return "%.3f\t%.3f\t%s\n" % (seg.start_seconds, seg.end_seconds, seg.text)


class TextWriter(_Writer):
    ext = "txt"

    def line(self, seg):
        return "[%s --> %s] %s\n" % (_clock(seg.start_seconds), _clock(seg.end_seconds), seg.text)


_BY_EXT = {cls.ext: cls for cls in (VTTWriter, SRTWriter, ASSWriter, JSONWriter, TSVWriter)}


def get_writer(fp, ext=None):
    if ext is None:
        ext = os.path.splitext(getattr(fp, "name", ""))[-1]   # keeps the dot: reference quirk
    return _BY_EXT.get(ext, TextWriter)(fp)
