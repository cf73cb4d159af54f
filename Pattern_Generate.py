"""Top-level drop-in name of the reference's pattern generator: `python Pattern_Generate.py -lj <path> -vctk <path> ...`
(reference Pattern_Generate.py:277-404; the implementation lives in multi_speaker_tts_amd/Pattern_Generate.py)."""
from multi_speaker_tts_amd.Pattern_Generate import *          # noqa: F401,F403
from multi_speaker_tts_amd.Pattern_Generate import main

if __name__ == "__main__":
    main()
