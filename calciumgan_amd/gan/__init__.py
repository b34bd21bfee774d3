"""Host-side mirror of the reference's ``gan`` package for the hot path
(gan/models, gan/algorithms) -- same registry names and object surface."""
