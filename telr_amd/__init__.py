"""telr_amd — MI355X-native alignment engine behind the TELR command line."""
__version__ = "0.1.0"
