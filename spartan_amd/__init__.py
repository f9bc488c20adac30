"""spartan_amd: an MI355X-native tile-execution backend behind the Spartan
expression-builder API."""
__version__ = '0.1.0'
