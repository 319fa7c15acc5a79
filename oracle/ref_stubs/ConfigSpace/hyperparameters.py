"""Empty stub."""
