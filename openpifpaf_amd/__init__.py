"""MI355X-native CifCaf decode path."""
