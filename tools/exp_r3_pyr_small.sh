#!/bin/bash
# round 3 experiment (development builds with a VSA_EXP_PYR_SMALL knob; the knob is not in the library): half-size top tiles for the pyramid launches.
#   pass 1 (frames up to 640 x 480 only; bit 0: <1 step, 16 x 16> for <1, 32 x 32>; bit 1: <4 steps, 2 x 2> for <4, 4 x 4>), configs[0]:
#       default 80.7 / 80.4 k   knob 1: 81.2 k   knob 2: 82.8 k   knob 3: 83.3 k
#   pass 2 (any frame size: half tiles whenever the launch would have fewer than 256 workgroups), two runs each:
#       configs[0] 80.7 / 80.6 -> 83.5 / 83.1 k   configs[1] 39.5 / 39.3 -> 38.8 / 38.8 k   configs[2] 10.18 / 10.17 -> 10.21 / 10.23 k   configs[4] 35.4 / 35.0 -> 34.0 / 34.2 k
# Kept: the rule by frame size (image_ops.hip, launch_pyramid_levels).  This script is the record; the knob builds are not in the tree.
echo "see the header"
