"""TensorBoard event writer (cleanba_amd.tb): CRC32C known answers, TFRecord framing, scalar round trip."""
import glob
import os

from cleanba_amd import tb


def test_crc32c_known_answers():
    assert tb.crc32c(b"123456789") == 0xE3069283          # iSCSI / RFC 3720 check value
    assert tb.crc32c(b"") == 0
    assert tb.crc32c(bytes(32)) == 0x8A9136AA             # RFC 3720 B.4: 32 bytes of zeros


def test_event_file_round_trip(tmp_path):
    w = tb.SummaryWriter(str(tmp_path))
    w.add_text("hyperparameters", "|param|value|")
    for i in range(5):
        w.add_scalar("charts/SPS", 1000.0 + i, i * 15360)
        w.add_scalar("losses/value_loss", 0.5 / (i + 1), i * 15360)
    w.close()
    files = glob.glob(os.path.join(str(tmp_path), "events.out.tfevents.*"))
    assert len(files) == 1
    got = tb.read_scalars(files[0])
    assert [g for g in got if g[1] == "charts/SPS"] == [(i * 15360, "charts/SPS", 1000.0 + i) for i in range(5)]
    assert len([g for g in got if g[1] == "losses/value_loss"]) == 5
    # first record is the file-version event TensorBoard requires
    with open(files[0], "rb") as f:
        assert b"brain.Event:2" in f.read(64)
