"""pypore_amd -- MI355X-native SpeedyStatSplit / FastStatSplit segmenter behind PyPore's
parser plug-in API.  `from pypore_amd.parsers import SpeedyStatSplit` is the drop-in for
`from PyPore.parsers import SpeedyStatSplit`."""
__version__ = "0.1.0"
