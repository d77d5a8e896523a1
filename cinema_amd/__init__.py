"""cinema_amd: MI355X-native (gfx950) compute path for the CineMA MAE hot path."""
