__version__ = "0.8.7+mi355x.1"
