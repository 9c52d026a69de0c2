"""Cut the --start-time / --duration fixtures from the REFERENCE's own code (run in the build container only).

    python tests/golden/make_time_fixtures.py [/root/reference]

`video.video_info.VideoInfo.time_to_frame` / `.validate_frame_range` are what the reference's CLI resolves
`--start-time`, `--duration`, `--start-frame`, `--frames` through (flow_processor.py:667-677, :1403-1420;
video/frame_extractor.py:88-98).  OpenCV is absent here: the module only needs the NAME cv2 at import time (an
empty stand-in module), and the video's properties are placed in VideoInfo's own cache instead of being probed
from a file.  Output: tests/golden/time_ranges.json (inputs + the reference's answers / error strings).
Nothing of the reference's source text is stored.  The GPU box never runs this file."""
import json
import os
import sys
import tempfile
import types

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    from video.video_info import VideoInfo
    dummy = tempfile.NamedTemporaryFile(suffix=".mp4", delete=False)
    dummy.close()
    cases = []
    for fps in (30.0, 24.0, 23.976, 29.97, 60.0, 12.5):
        for total in (1, 7, 300, 1801):
            vi = VideoInfo(dummy.name)
            vi._info_cache = {'fps': fps, 'width': 64, 'height': 64, 'total_frames': total,
                              'duration_seconds': total / fps, 'path': dummy.name}
            for start_time, duration, start_frame, frames in (
                    (None, None, 0, 1000), (None, None, 5, 3), (0.0, 1.0, 0, 1000), (0.5, None, 0, 10),
                    (None, 0.25, 2, 1000), (1.999, 0.1, 0, 1000), (2.0, 2.0, 9, 9), (10.0, 1.0, 0, 5),
                    (0.0333, 0.0667, 0, 5), (100.0, 1.0, 0, 5), (None, None, -3, 4), (None, None, 5000, 4)):
                s, n = start_frame, frames
                rec = {"fps": fps, "total": total, "start_time": start_time, "duration": duration,
                       "start_frame": start_frame, "frames": frames}
                if start_time is not None:
                    s = vi.time_to_frame(start_time)
                if duration is not None:
                    n = vi.time_to_frame(duration)
                rec["resolved"] = [s, n]
                try:
                    rec["range"] = list(vi.validate_frame_range(s, n))
                except ValueError as e:
                    rec["error"] = str(e)
                cases.append(rec)
    os.unlink(dummy.name)
    with open(os.path.join(HERE, "time_ranges.json"), "w") as f:
        json.dump({"cases": cases}, f, indent=0)
    print(f"{len(cases)} cases")


if __name__ == "__main__":
    main()
