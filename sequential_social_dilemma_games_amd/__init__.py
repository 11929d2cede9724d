"""MI355X-native vectorised engine for the Harvest / Cleanup social-dilemma gridworlds."""
