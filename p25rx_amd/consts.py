"""Constants of the reference (src/consts.rs:4-13)."""
BUF_COUNT = 1                      # src/consts.rs:4
BUF_BYTES = 32768                  # src/consts.rs:6
BUF_SAMPLES = BUF_BYTES // 2       # src/consts.rs:8
SDR_SAMPLE_RATE = 240000           # src/consts.rs:11
BASEBAND_SAMPLE_RATE = 48000       # src/consts.rs:13
