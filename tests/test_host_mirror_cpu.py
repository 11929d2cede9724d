"""Host-side helpers of the dict mirror that need no GPU."""
import numpy as np
import pytest

from sequential_social_dilemma_games_amd.map_env import MapEnv

# the reference's DEFAULT_COLOURS (social_dilemmas/envs/map_env.py:24-41) has the key '' next to the one-character glyphs
REFERENCE_STYLE_COLOURS = {' ': [0, 0, 0], '0': [0, 0, 0], '': [180, 180, 180], '@': [180, 180, 180], 'A': [0, 255, 0],
                           'F': [255, 255, 0], 'P': [159, 67, 255], '1': [159, 67, 255], '2': [2, 81, 154]}


def test_map_to_colors_takes_the_references_colour_dict():
    """a caller who passes the reference's dict -- with its '' key -- gets the reference's cell-by-cell result,
    not a TypeError from ord('')."""
    grid = np.array([['@', 'A', ' '], ['1', '', '2']], dtype='<U1')
    got = MapEnv.map_to_colors(None, grid, REFERENCE_STYLE_COLOURS)
    want = np.array([[REFERENCE_STYLE_COLOURS[c] for c in row] for row in grid.tolist()])
    np.testing.assert_array_equal(got, want)
    assert got.shape == (2, 3, 3)
    # keys that can never equal a '<U1' cell are ignored, a glyph without a colour is a KeyError as in the reference
    MapEnv.map_to_colors(None, grid, dict(REFERENCE_STYLE_COLOURS, **{'AB': [1, 2, 3]}))
    with pytest.raises(KeyError):
        MapEnv.map_to_colors(None, np.array([['Z']], dtype='<U1'), REFERENCE_STYLE_COLOURS)
