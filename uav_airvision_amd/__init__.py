"""MI355X-native implementation of UAV-Airvision's per-frame hot path."""
