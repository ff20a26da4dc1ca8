"""CPU tests of the I/O restatements on the committed dataset excerpts (no GPU): the XCP loader's
skip rule and derived cameras, the CSV reader's stoi/gap semantics, PLY formatting."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_xcp_loader_skips_cameras_without_control_frame(oracle):
    import io_oracle as io
    import json
    cams = io.load_cameras_xml(os.path.join(GOLD, "r02_d1_excerpt", "cameras.xcp"))
    assert [c["id"] for c in cams] == [55260362, 60881257, 61016180, 61920488]  # the 2 IR cameras are skipped
    g = json.load(open(os.path.join(GOLD, "r02_d1.json")))
    np.testing.assert_allclose(np.stack([c["P"].reshape(12) for c in cams]), g["expected"]["literal"]["P"],
                               rtol=1e-12, atol=1e-9)
    fix = io.load_cameras_xml(os.path.join(GOLD, "r02_d1_excerpt", "cameras.xcp"), vicon_order=True)
    np.testing.assert_allclose(np.stack([c["P"].reshape(12) for c in fix]), g["expected"]["fixup"]["P"],
                               rtol=1e-12, atol=1e-9)
    assert all(c["width"] == 1924 and c["height"] == 1082 for c in cams)


def test_csv_reader_semantics(oracle, tmp_path):
    import io_oracle as io
    import json
    counts, dets = io.read_detection_csvs(os.path.join(GOLD, "r02_d1_excerpt", "dl_data"))
    assert counts.shape == (60, 4) and (counts == 1).all()
    g = json.load(open(os.path.join(GOLD, "r02_d1.json")))
    first = dets.reshape(60, 4, 2)[:48]
    ref = np.array([g["detections"][str(c["DEVICEID"])][:48] for c in g["cameras"]]).transpose(1, 0, 2)
    np.testing.assert_array_equal(first, ref)
    counts4, dets4 = io.read_detection_csvs(os.path.join(GOLD, "r12_d4_excerpt", "dl_data"))
    assert counts4.shape == (40, 4) and counts4.max() >= 3 and dets4.shape[0] == counts4.sum()
    # float fields are truncated by stoi, missing frame numbers become empty frames, bad records throw
    d = tmp_path / "dl"
    d.mkdir()
    (d / "a.csv").write_text("0,1.9,2,3,4,10.7,20.2,0.9\n3,1,2,3,4,11,21,0.9,5,6,7,8,12,22,0.8\n")
    (d / "b.csv").write_text("0,1,2,3,4,30,40,0.9\n1,1,2,3,4,31,41,0.9\n2,1,2,3,4,32,42,0.9\n3,1,2,3,4,33,43,0.9\n")
    c, dd = io.read_detection_csvs(str(d))
    assert c.tolist() == [[1, 1], [0, 1], [0, 1], [2, 1]]
    assert dd[0].tolist() == [10.0, 20.0] and dd[-3].tolist() == [11.0, 21.0]
    (d / "b.csv").write_text("0,1,2,3,4,30,40\n")
    with pytest.raises(RuntimeError, match="Invalid CSV file!"):
        io.read_detection_csvs(str(d))


def test_ply_text_formatting():
    import io_oracle as io
    t = io.ply_text(np.array([[1270.280557, -975.7791806, 0.0], [1e-5, 123456789.0, -0.5]]))
    assert t.splitlines()[2] == "element vertex 2"
    assert t.splitlines()[-2:] == ["1270.28 -975.779 0", "1e-05 1.23457e+08 -0.5"]
