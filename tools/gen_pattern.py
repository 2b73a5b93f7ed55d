#!/usr/bin/env python3
"""Regenerate the rBRIEF sampling-pattern data table from the reference.

The table is DATA (the 256 learned test pairs of the ORB paper, as carried by
the reference at src/ORBextractor.cc:148-406).  This script only runs where
/root/reference exists; the emitted .inc files are committed.  Format is our
own: 1024 signed bytes, 16 per line (= 4 test pairs), so one line == 4 bits.
Known-answer check: sha256 over the int32-LE encoding, sum, range.
"""
import re, hashlib, struct, sys, os

REF = "/root/reference/src/ORBextractor.cc"
SHA = "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"

def main():
    lines = open(REF).read().split("\n")[147:406]
    txt = re.sub(r"/\*.*?\*/", "", "\n".join(lines))
    nums = [int(x) for x in re.findall(r"-?\d+", txt.split("=", 1)[1])]
    assert len(nums) == 1024 and sum(nums) == -406
    assert hashlib.sha256(struct.pack("<1024i", *nums)).hexdigest() == SHA
    body = ["/* rBRIEF pattern: 256 x (x0,y0,x1,y1), int8. sha256(int32-LE)=%s */" % SHA]
    for i in range(0, 1024, 16):
        body.append(",".join("%d" % v for v in nums[i:i + 16]) + ",")
    out = "\n".join(body) + "\n"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in ("oracle/orb_pattern.inc", "orb-slam3-mac_amd/csrc/orb_pattern.inc"):
        open(os.path.join(root, p), "w").write(out)
    print("ok")

if __name__ == "__main__":
    main()
