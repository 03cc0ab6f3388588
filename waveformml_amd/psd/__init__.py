"""Host-side mirror of the reference's PSD training interface (LitPSD / SPConvNet / collate / config
plugin loader) for the MI355X sparse-conv path.  See DESIGN.md for the reference file:line map."""
