"""USAGE

    reazonspeech-espnet-asr [-h] [--to={vtt,srt,ass,json,tsv}] [-o file] audio

Command line front of `transcribe()` (pkg/espnet-asr/src/cli.py: the nemo package's cli with this package's functions).
"""
import getopt
import sys
import warnings


def main(argv=None):
    from .writer import get_writer
    from .audio import audio_from_path
    from .transcribe import transcribe, load_model
    opts, args = getopt.getopt(sys.argv[1:] if argv is None else argv, "ho:", ("help", "output=", "to="))
    outpath = fmt = None
    for key, val in opts:
        if key in ("-h", "--help"):
            print(__doc__, file=sys.stderr)
            return None
        if key in ("-o", "--output"):
            outpath = val
        elif key == "--to":
            fmt = val
    if not args:
        print("no audio file specified", file=sys.stderr)
        print(__doc__, file=sys.stderr)
        return 1
    warnings.simplefilter("ignore")
    result = transcribe(load_model(), audio_from_path(args[0]))
    out = open(outpath, "w") if outpath is not None else sys.stdout
    with out:
        writer = get_writer(out, fmt)
        writer.write_header()
        for seg in result.segments:
            writer.write(seg)
    return None


if __name__ == "__main__":
    sys.exit(main())
